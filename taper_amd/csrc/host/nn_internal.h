// nn_internal.h -- what modules.cpp / optim.cpp / comm.cpp / trainer.cpp share beyond taper.h (host mirror internals; not installed).
#pragma once
#include "taper.h"

#define TH(call) th_check((call), #call)

namespace taper {
// modules.cpp, used by the Trainer's choice of step form (trainer.cpp)
size_t mlp2_min_batch();                                                        // TAPER_MLP2_MIN_BATCH: from this batch on th_mlp2_xent
bool mlp2_shapes_ok(size_t batch, const std::vector<Tensor> &w, int64_t n_rows);
bool mlp3_fuse();                                                               // TAPER_MLP3: the three-layer classifier as two launches
}  // namespace taper
