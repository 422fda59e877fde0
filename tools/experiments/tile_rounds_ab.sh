#!/bin/bash
# usage (on the GPU box, from the repo root): tools/experiments/tile_rounds_ab.sh > gpurun_out/tile_rounds_ab.txt
# Does the slot kernel's time follow ROUNDS of workgroups (2048 resident at 8 per CU)?  The 256 x 2 tile against 256 x 3 / 256 x 4 builds
# (libchub_t3.so / libchub_t4.so: make KFLAGS="-DCHUB_BIG_BLOCK=256 -DCHUB_BIG_SLOTS_PER_LANE=3 -DCHUB_XCD_ANY_TILE=1", --tile large) on the
# [20, 25] hub at batch sizes whose workgroup counts fall on different fractions of a round, a b c a b c inside one call.
P=$PWD/charginghub-env_amd
for n in 32768 24576 40960 49152 65536; do
  for rep in 1 2; do
    python3 tools/ab_step_times.py --config c4 --envs $n --tile auto
    python3 tools/ab_step_times.py --config c4 --envs $n --tile large --lib $P/libchub_t3.so
    python3 tools/ab_step_times.py --config c4 --envs $n --tile large --lib $P/libchub_t4.so
  done
done
