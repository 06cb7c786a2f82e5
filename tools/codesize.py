#!/usr/bin/env python3
"""Instruction count per source line of one kernel (development tool): hipcc -g -save-temps, parse .loc."""
import collections, re, subprocess, sys, os, tempfile
kern = sys.argv[1] if len(sys.argv) > 1 else "_Z16bg_engine_kernelILb0ELb0ELb0E"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
d = tempfile.mkdtemp()
subprocess.check_call(["hipcc", "-O3", "-g", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                       "-Wno-unused-value", "-save-temps=obj", "-o", d + "/l.so",
                       os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "balatro_gym_amd/csrc/bg_lib.hip")],
                      stderr=subprocess.DEVNULL)
s = open(d + "/bg_lib-hip-amdgcn-amd-amdhsa-gfx950.s").read()
m = re.search(r'^' + re.escape(kern) + r'.*?:\n(.*?)\n\s+s_endpgm', s, re.S | re.M)
files = {}
for fm in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    files[int(fm.group(1))] = (fm.group(3) or fm.group(2)).split('/')[-1]
cur = (0, 0); cnt = collections.Counter(); total = 0
for line in m.group(1).split('\n'):
    lm = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', line)
    if lm:
        cur = (int(lm.group(1)), int(lm.group(2))); continue
    if re.match(r'\s+(v_|s_|global_|ds_|buffer_|scratch_|flat_)', line):
        cnt[cur] += 1; total += 1
print("total instrs", total)
for (f, l), c in cnt.most_common(top):
    print(f"{files.get(f, f)}:{l}  {c}")
