#!/usr/bin/env python3
"""Development aid: cycle counters of the step engine kernel (build with -DBG_TIMING4: tools/build_variant.sh t4 -DBG_TIMING4,
then BALATRO_MI355X_LIB=build/variants/t4.so python tools/timing4.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv, _native as nat
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n, T = int(os.environ.get("N", "65536")), int(os.environ.get("T", "372"))
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
WARM = int(os.environ.get("WARM", str(T)))  # steps per warm-up launch (372: the bench's machine state before a short launch)
rb = RowBuffers(n, env.device, steps=max(T, WARM))
NW = int(os.environ.get("NWARM", "3"))
for i in range(NW):
    env.rollout(WARM, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * WARM, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
L = nat.load()
out = (C.c_ulonglong * 32)()
L.bg_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
L.bg_debug_counters(env._h, out)
env.set_profiling(True)
env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=NW * WARM, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
p = env.get_profile()
L.bg_debug_counters(env._h, out)
o = list(out)
waves = max(1, o[1]); wgs = n / 256  # o[1] counts WORKER waves
print(f"launch {p['rollout_ms']*1e3:.0f} us, waves {waves}, T {T}; wave cycles {o[0]/waves:.0f}, idle {o[11]/max(1,o[0]):.2f} of it, failed claims per wave {o[12]/waves:.1f}")
tot_busy = 0
for c, nm in ((0, "run"), (1, "play"), (2, "other")):
    b = max(1, o[2 + 3 * c])
    tot_busy += o[4 + 3 * c]
    print(f"  {nm:5s}: batches per workgroup-step {o[2+3*c]/wgs/T:.2f}  items per batch {o[3+3*c]/b:.1f}  cycles per batch {o[4+3*c]/b:.0f}  share of wave time {o[4+3*c]/max(1,o[0]):.2f}")
print(f"  claims: successful {o[14]/max(1,o[0]):.3f} of wave time ({o[14]/max(1,o[2]+o[5]+o[8]):.0f} cycles each), failed {o[15]/max(1,o[0]):.3f} ({o[15]/max(1,o[12]):.0f} cycles each)")
print(f"  copy-out: {o[13]/max(1,o[2]+o[5]+o[8]):.0f} cycles per batch, {o[13]/max(1,o[0]):.2f} of wave time")
print(f"  batches of <= 4 items: play {o[17]/wgs:.2f} per workgroup at {o[18]/max(1,o[17]):.0f} cycles, other {o[19]/wgs:.2f} at {o[30]/max(1,o[19]):.0f} cycles")
rb_ = max(1, o[2])
print(f"  run batch (lane 0's view; cycles per batch): item fetch {o[26]/rb_:.0f}, cheap step {o[27]/rb_:.0f}, finish {o[28]/rb_:.0f}, copy-out (all classes) {o[13]/max(1,o[2]+o[5]+o[8]):.0f}, push back + further steps {o[29]/rb_:.0f}")
wg = n / 256
if o[20]:
    t0w = (~o[20]) & 0xFFFFFFFFFFFFFFFF
    print(f"  workgroup timeline (us): prologue {o[16]/wg/2400:.1f} (cycles/2400) | half of its envs through {o[24]/wg/100:.1f}, 15/16 {o[25]/wg/100:.1f}, all {o[22]/wg/100:.1f} (mean over workgroups), slowest workgroup {o[23]/100:.1f}; first start -> last end {(o[21]-t0w)/100:.1f}")
env.close()
