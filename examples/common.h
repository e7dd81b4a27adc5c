// Shared by the two example drivers: dataset selection and flag parsing.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "../taper_amd/csrc/host/taper.h"

namespace ex {

struct Args {
    std::string data_dir = "data/mnist";  // data/mnist.rs:13 default
    size_t epochs = 0 /* 0 = the example's own default */, batch_size = 256, train_n = 60000, test_n = 10000;
    bool eager = false;                   // --eager: the literal per-step loop of the reference example
};

inline Args parse(int argc, char **argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        auto val = [&](const char *flag) { return (!strcmp(argv[i], flag) && i + 1 < argc) ? argv[++i] : nullptr; };
        if (const char *v = val("--data-dir")) a.data_dir = v;
        else if (const char *v = val("--epochs")) a.epochs = strtoul(v, nullptr, 10);
        else if (const char *v = val("--batch-size")) a.batch_size = strtoul(v, nullptr, 10);
        else if (const char *v = val("--train-n")) a.train_n = strtoul(v, nullptr, 10);
        else if (const char *v = val("--test-n")) a.test_n = strtoul(v, nullptr, 10);
        else if (!strcmp(argv[i], "--eager")) a.eager = true;
        else {
            fprintf(stderr, "usage: %s [--data-dir D] [--epochs N] [--batch-size B] [--train-n N] [--test-n N] [--eager]\n", argv[0]);
            exit(2);
        }
    }
    return a;
}

inline bool exists(const std::string &p) { return std::ifstream(p).good(); }

// MNISTDataset::new (data/mnist.rs:165-183) without the downloader: the IDX files when both are
// on disk, otherwise SURVEY 8(d)'s synthetic rows (this checkout ships labels only).
inline taper::MNISTDataset load(const Args &a, bool train) {
    const std::string img = a.data_dir + (train ? "/train_images" : "/test_images");
    const std::string lab = a.data_dir + (train ? "/train_labels" : "/test_labels");
    if (exists(img) && exists(lab)) return taper::MNISTDataset::from_idx_files(img, lab, train);
    printf("   (%s not found: synthetic %s set)\n", img.c_str(), train ? "training" : "test");
    return taper::MNISTDataset::synthetic(train ? a.train_n : a.test_n, train ? 0x7461706572ull : 0x7461706573ull, train);
}

}  // namespace ex
