# A/B inside one GPU call: libchub_a.so (baseline build) against the tree's libchub.so; AB_WHAT selects the sizes
mkdir -p gpurun_out/$1
A=${AB_LIB:-$PWD/charginghub-env_amd/libchub_a.so}
for rep in 1 2; do
  for cfg in ${AB_WHAT:-c4 c5}; do
    case $cfg in
      c4) E="--config c4";; c5) E="--config c5";; c2) E="--config c2";; *) E="--envs $cfg";;
    esac
    python3 tools/ab_step_times.py $E --lib $A; python3 tools/ab_step_times.py $E
  done
done > gpurun_out/$1/ab.log 2>&1
cat gpurun_out/$1/ab.log
