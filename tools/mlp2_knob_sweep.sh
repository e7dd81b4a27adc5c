#!/bin/bash
# Runs on the GPU box: the large-batch MLP step (th_mlp2_xent) at 4 096 and 1 024 rows under each of its launch-shape knobs (TAPER_MLP2_RT / _NW /
# _KSPLIT / _KZ), one bench.py --batch run each: are the defaults still the optimum?  (r06: yes.)
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --gpus 1 --batch $B --steps 600 --warmup 60 --workloads none --no-sweep --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2))"; }
for B in 4096 1024; do
  echo "B=$B default: $(run X=1)"
  for rt in 16 32; do for nw in 4 8; do echo "B=$B RT=$rt NW=$nw: $(run TAPER_MLP2_RT=$rt TAPER_MLP2_NW=$nw)"; done; done
  echo "B=$B KSPLIT=2: $(run TAPER_MLP2_KSPLIT=2)  KSPLIT=4: $(run TAPER_MLP2_KSPLIT=4)"
  echo "B=$B KZ=18: $(run TAPER_MLP2_KZ=18) KZ=24: $(run TAPER_MLP2_KZ=24) KZ=30: $(run TAPER_MLP2_KZ=30)"
done
