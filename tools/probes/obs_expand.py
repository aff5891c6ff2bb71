"""Diagnostic driver of obs_expand.hip: rate of the observation block of simple_spread 64/64 written by one-store
workgroups from a per-step record, alone and beside the step kernel of a library built without observation rows.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o variants/libobsx.so tools/probes/obs_expand.hip
    tools/build_variant.sh noobs mpe_wave.hip -DMPE_ABLATE_OBS
    MPE_HIP_LIB=$PWD/variants/libmpe_noobs.so python3 tools/probes/obs_expand.py [chunk steps ...]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multiagent-particle-envs_amd"))
import mpe_hip  # noqa: E402

A = L = 64
RP = 2 + L + A - 1          # coordinate pairs per observation row
QW = A * RP // 2            # 16-byte quads per world-step
REC, ZERO = 192, 192
B, T = 2048, 25


def table():
    idx = np.zeros((A * RP, 2), np.uint32)
    for i in range(A):
        for r in range(RP):
            if r == 0:
                first, sub = 128 + i, ZERO
            elif r == 1:
                first, sub = i, ZERO
            else:
                m = r - 2 - L
                first, sub = (A + r - 2 if m < 0 else m + (1 if m >= i else 0)), i
            idx[i * RP + r] = (first, sub)
    q = idx.reshape(QW, 4)
    return (q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16) | (q[:, 3] << 24)).astype(np.uint32)


def expected(rec_ws):
    """One world-step's block from its record [192, 2]: the layout of simple_spread.observation."""
    P, V = rec_ws[:128], rec_ws[128:]
    rows = []
    for i in range(A):
        others = np.delete(P[:A], i, axis=0) - P[i]
        rows.append(np.concatenate([V[i:i + 1], P[i:i + 1], P[A:] - P[i], others]).reshape(-1))
    return np.stack(rows)


def timed(fn, n=10, stream=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


def main():
    chunks = [int(x) for x in sys.argv[1:]] or [5]
    dev = torch.device("cuda:0")
    lib = C.CDLL(os.path.join(ROOT, "variants", "libobsx.so"))
    lib.obsx_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
    tab = torch.from_numpy(table().view(np.int32)).to(dev)
    rec = torch.randn(T * B, REC, 2, device=dev)
    obs = torch.empty(T, B, A * 2 * RP, device=dev)
    nbytes = obs.numel() * 4

    def launch_x(variant, t0=0, nt=T, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream()
        err = lib.obsx_launch(variant, rec[t0 * B:].data_ptr(), tab.data_ptr(), obs[t0:].data_ptr(), nt * B, C.c_void_p(st.cuda_stream))
        assert err == 0, err

    want = expected(rec[5].cpu().numpy())
    for variant in (0, 1, 3):
        obs.zero_()
        launch_x(variant)
        torch.cuda.synchronize()
        got = obs.view(T * B, A, 2 * RP)[5].cpu().numpy()
        assert np.array_equal(got, want), variant
    for variant in (0, 1, 2, 3):
        us = timed(lambda: launch_x(variant))
        print("expansion alone, variant %d: %.1f us, %.2f TB/s" % (variant, us, nbytes / us / 1e6), flush=True)

    env = mpe_hip.BatchedEnv("simple_spread", B, n_agents=A, n_landmarks=L, dtype=torch.float32, device="cuda:0")
    gen = torch.Generator(dev).manual_seed(1)
    act = torch.randint(0, 5, (T, B, A), dtype=torch.int32, device=dev, generator=gen)
    out = env.rollout(act, reset_seed=3, obs0=True)
    us = timed(lambda: env.rollout(act, out=out, reset_seed=3, obs0=True))
    print("step kernel alone (library %s): %.1f us per 25-step episode launch" % (os.environ.get("MPE_HIP_LIB", "in-tree"), us), flush=True)

    s_x = torch.cuda.Stream(dev)
    for ch in chunks:
        n_ch = (T + ch - 1) // ch
        outs = []
        for c in range(n_ch):
            t0, t1 = c * ch, min(T, (c + 1) * ch)
            o = {k: (v[t0:t1] if v is not None and k != "obs0" else v) for k, v in out.items()}
            o["obs0"] = out["obs0"] if c == 0 else None
            outs.append((t0, t1, o))
        ev = [torch.cuda.Event() for _ in range(n_ch)]
        done = torch.cuda.Event()

        def pipeline(variant=0):
            main_s = torch.cuda.current_stream()
            for c, (t0, t1, o) in enumerate(outs):
                if c == 0:
                    env.rollout(act[t0:t1], out=o, reset_seed=3, obs0=True)
                else:
                    env.rollout(act[t0:t1], out=o)
                ev[c].record(main_s)
                s_x.wait_event(ev[c])
                launch_x(variant, t0, t1 - t0, s_x)
            done.record(s_x)
            main_s.wait_event(done)

        for variant in (0, 3):
            us = timed(lambda: pipeline(variant))
            print("pipeline, %d-step chunks, expansion variant %d: %.1f us per episode (%.3f of 8 TB/s on %d B)"
                  % (ch, variant, us, (nbytes + T * B * (A * 4 + 1 + 1.25 * A * 4)) / us / 1e6 / 8, nbytes), flush=True)
        us = timed(lambda: [env.rollout(act[t0:t1], out=o) for t0, t1, o in outs])
        print("step kernel alone in %d-step chunks: %.1f us" % (ch, us), flush=True)


if __name__ == "__main__":
    main()
