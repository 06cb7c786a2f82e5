import ctypes as C, os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd())
sys.path.insert(0, os.getcwd())
import torch
from balatro_gym_amd import BalatroVecEnv, _native as nat
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n = 65536
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=4, row_stride=384)
L = nat.load(); out = (C.c_ulonglong * 32)()
L.bg_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
L.bg_debug_counters(env._h, out)
for T in (1, 1, 1):
    env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, obs_buffers=rb, zero_stats=False)
    torch.cuda.synchronize(); L.bg_debug_counters(env._h, out); o = list(out)
    for c, nm in ((0, "run"), (1, "play"), (2, "other")):
        b = max(1, o[2 + 3 * c])
        print(f"  {nm:5s}: batches per workgroup {o[2+3*c]/256:.2f}  items per batch {o[3+3*c]/b:.1f}  cycles per batch {o[4+3*c]/b:.0f}")
    print("--")
