# Bench lines for every BASELINE.json config (per-GPU share) plus the batch-size sweep of config 2.
set -e
mkdir -p gpurun_out/configs
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs --no-sustained "$@" > gpurun_out/configs/$name.json 2>/dev/null; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]; d=json.load(open('gpurun_out/configs/%s.json'%n))
r=d['roofline']; s=d.get('step_mode') or {}
print('%-22s %7.2f G agent-steps/s  %.3f us/step | kernel %.4f ms  %.0f GB/s frac %.3f | step-mode %.2f G/s %.2f us frac %.3f' % (n, d['value']/1e9, d['ms_per_step']*1e3, r['launch_ms'], r['achieved'], r['frac'], s.get('value',0)/1e9, s.get('launch_ms_hip_events',0)*1e3, s.get('roofline_frac',0)))
PY
}
run spread3_B65536
run spread3_B262144 --envs 262144 --steps 1000 --warmup 100
run spread3_B1048576 --envs 1048576 --steps 500 --warmup 50
run spread3_B4194304 --envs 4194304 --steps 100 --warmup 25 --no-step-mode
run spread3_f64_B65536 --dtype f64 --steps 500 --warmup 50
run tag_B131072 --scenario simple_tag --envs 131072 --steps 1000 --warmup 100
run wcomm_B262144 --scenario simple_world_comm --envs 262144 --steps 500 --warmup 50
run spread64_B2048 --agents 64 --landmarks 64 --envs 2048 --steps 500 --warmup 50
run spread64_f64_B2048 --agents 64 --landmarks 64 --envs 2048 --dtype f64 --steps 100 --warmup 25
run simple_B262144 --scenario simple --envs 262144 --steps 1000 --warmup 100
run adversary_B262144 --scenario simple_adversary --envs 262144 --steps 1000 --warmup 100
run push_B262144 --scenario simple_push --envs 262144 --steps 1000 --warmup 100
run speaker_listener_B262144 --scenario simple_speaker_listener --envs 262144 --steps 1000 --warmup 100
run reference_B262144 --scenario simple_reference --envs 262144 --steps 1000 --warmup 100
run crypto_B262144 --scenario simple_crypto --envs 262144 --steps 1000 --warmup 100
run predator_prey_B131072 --scenario simple_predator_prey --envs 131072 --steps 500 --warmup 50
run consensus_B65536 --scenario consensus --envs 65536 --steps 500 --warmup 50
run collector_B65536 --scenario collector --envs 65536 --steps 250 --warmup 50
