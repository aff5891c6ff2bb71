import sys, os, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multiagent-particle-envs_amd"))
import torch, mpe_hip, bench
dev = torch.device("cuda:0")
def rec(tag, B=2048, scenario="simple_spread", kw=dict(n_agents=64, n_landmarks=64)):
    r = bench.config_record(torch, mpe_hip, dev, tag, scenario, B, "f32", kw)
    print("%-50s %.4f ms frac %.3f" % (tag, r["launch_ms"], r["frac"]), flush=True)
rec("spread64 fresh")
rec("spread64 again")
rec("1M fresh-ish", 1048576, "simple_spread", {})
# heat: headline sustained 3 s
env = mpe_hip.BatchedEnv("simple_spread", 65536, dtype=torch.float32, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(1)
a = bench.random_actions(torch, env, 25, gen)
ms, wall, n = bench.time_episode_launches(torch, env, a, 3.0, 0.0)
print("headline 3 s: %.4f ms" % ms, flush=True)
rec("spread64 after 3 s of headline launches")
rec("1M after", 1048576, "simple_spread", {})
del env, a
torch.cuda.empty_cache()
rec("spread64 after empty_cache")
time.sleep(5)
rec("spread64 after 5 s idle")
x = torch.empty(123456789, dtype=torch.uint8, device=dev)   # shifts later allocations
rec("spread64 with a 123 MB odd-sized block before it")
rec("tag", 131072, "simple_tag", {})
rec("wcomm", 262144, "simple_world_comm", {})
rec("spread64 after tag and wcomm")
