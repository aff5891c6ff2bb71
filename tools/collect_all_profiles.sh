#!/bin/bash
# On the GPU box: tools/collect_profiles.py for the eight profiled configs, one after the other, into
# gpurun_out/<tag>/prof_<name>/.  tools/merge_profiles.py then copies the summaries into profiles/.
# usage: bash tools/collect_all_profiles.sh r02
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; echo "== $name"; timeout -k 10 900 python3 tools/collect_profiles.py --out gpurun_out/$tag/prof_$name "$@" > gpurun_out/$tag/prof_$name.log 2>&1 || { tail -5 gpurun_out/$tag/prof_$name.log; exit 1; }; tail -2 gpurun_out/$tag/prof_$name.log; }
mkdir -p gpurun_out/$tag
run spread3 &&
run tag --scenario simple_tag --envs 131072 &&
run wcomm --scenario simple_world_comm --envs 262144 &&
run spread64 --envs 2048 -- --agents 64 --landmarks 64 --no-step-mode &&   # (single steps run the same kernel: they would mix into its average)
run spread64_f64 --envs 2048 --dtype f64 -- --agents 64 --landmarks 64 --no-step-mode &&
run consensus --scenario consensus --envs 65536 &&
run collector --scenario collector --envs 65536 &&
run pp --scenario simple_predator_prey --envs 131072
