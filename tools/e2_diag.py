"""Development: run fused packed-record rollouts through the two-kernel engine at several sizes and dump its control block
(queue tails / heads, arrival counters, wall-clock stamps) -- how a stuck hand-over is localised."""
import ctypes as C
import sys
import time

import numpy as np
import torch

from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers


def dump(env, label):
    out = (C.c_uint32 * 4096)()
    env._L.bg_debug_e2.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    n = env._L.bg_debug_e2(env._h, C.cast(out, C.c_void_p), 4096)
    w = np.frombuffer(out, dtype=np.uint32)[:n]
    st = w[96:112].view(np.uint64)
    inv = lambda v: (~np.uint64(v)) if v else 0
    t_o0, t_o1, t_s0, t_s1 = int(inv(st[2])), int(st[3]), int(inv(st[4])), int(st[5])
    print(f"   first instruction: service - owner = {(int(inv(st[6])) - int(inv(st[7]))) / 100:.1f} us; owner prologue {(t_o0 - int(inv(st[7]))) / 100:.1f} us; service prologue {(t_s0 - int(inv(st[6]))) / 100:.1f} us")
    print(f"[{label}] owners_done={w[0]} svc_seen={w[64:72].tolist()} batches={st[0]} reqs={st[1]}"
          f" owner span={(t_o1 - t_o0) / 100:.1f} us svc start-owner start={(t_s0 - t_o0) / 100:.1f} us svc end-owner end={(t_s1 - t_o1) / 100:.1f} us")
    q = w[128:128 + 32 * 64].reshape(32, 64)
    pend = [(i, int(q[i, 0]), int(q[i, 32])) for i in range(32) if q[i, 0] != q[i, 32]]
    print("   queues with tail != head (queue, tail, head):", pend[:16])
    print("   tails per xcc:", [int(q[4 * x:4 * x + 4, 0].sum()) for x in range(8)])
    b = 128 + 32 * 64
    ex, why, dn, tg = w[b:b + 8], w[b + 8:b + 16], w[b + 16:b + 24], (~w[b + 24:b + 32])
    t0x = ~(w[b + 32:b + 48].view(np.uint64)); t1x = w[b + 48:b + 64].view(np.uint64); bat = w[b + 64:b + 80].view(np.uint64)
    tm = w[b + 120:b + 184].view(np.uint64) if len(w) >= b + 184 else None
    if tm is not None and tm.any():
        o, sv = tm[:12].astype(np.float64), tm[16:28].astype(np.float64)
        print(f"   OWNER per iteration (cycles; {o[10]:.0f} iterations, {o[11] / max(o[10], 1):.1f} steps finished per iteration): issue polls {o[0] / o[10]:.0f} cheap {o[1] / o[10]:.0f} post {o[2] / o[10]:.0f} account+copy {o[3] / o[10]:.0f} absorb (wait) {o[4] / o[10]:.0f} entries+issue loads {o[5] / o[10]:.0f}")
        print(f"   OWNER-side service step: entry written -> answer seen {tm[12] / max(tm[14], 1) / 100:.1f} us, answer seen -> absorbed {tm[13] / max(tm[14], 1) / 100:.1f} us ({tm[14]} service steps)")
        print(f"   SERVICE-side: entry written -> batch started {tm[28] / max(tm[29], 1) * 0.64:.1f} us")
        print(f"   SERVICE per batch (cycles; {sv[10]:.0f} batches of {sv[11] / max(sv[10], 1):.1f}): poll+claim {sv[0] / sv[10]:.0f} entries {sv[1] / sv[10]:.0f} state {sv[2] / sv[10]:.0f} dispatch {sv[3] / sv[10]:.0f} cap+reset {sv[4] / sv[10]:.0f} mask+pack+stores {sv[5] / sv[10]:.0f} record {sv[6] / sv[10]:.0f} drain {sv[7] / sv[10]:.0f} answer {sv[8] / sv[10]:.0f}; FAILED CLAIMS per batch {sv[9] / sv[10]:.2f}")
    print("   MIGRATED waves (owner, service):", w[b + 80:b + 82].tolist())
    print("   per xcc: exits", ex.tolist(), "reasons", why.tolist(), "done seen", dn.tolist(), "min target", tg.tolist())
    print("   per xcc: first leave - owner end (us)", [round((int(v) - t_o1) / 100, 1) for v in t0x], "last leave - owner end", [round((int(v) - t_o1) / 100, 1) for v in t1x], "batches", bat.tolist())


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384, 65536]
    T = int(__import__("os").environ.get("T", "24"))
    for n in sizes:
        from tests.test_gpu_parity import _vec
        env = _vec(n, [1000 + i for i in range(n)], autoreset=True, fused_steps=T)
        rb = RowBuffers(n, env.device, steps=T, row_stride=384)
        import contextlib, os
        strm = torch.cuda.Stream() if os.environ.get("STREAM") else None
        for rep in range(int(os.environ.get('REPS', '3'))):
            torch.cuda.synchronize()
            t0 = time.time()
            with (torch.cuda.stream(strm) if strm else contextlib.nullcontext()):
                env.rollout(T, policy=2, policy_seed=7 + rep, obs_buffers=rb)
            torch.cuda.synchronize()
            dt = time.time() - t0
            try:
                env.check()
                ok = "ok"
            except Exception as exc:  # noqa: BLE001
                ok = "FAILED: " + str(exc)[60:130]
            print(f"N={n} T={T} rep {rep}: {dt * 1e6:.0f} us {ok} stats={env._stats.cpu().numpy().tolist()}")
            dump(env, f"N={n} rep {rep}")
            if ok != "ok":
                break
        env.close()


if __name__ == "__main__":
    main()
