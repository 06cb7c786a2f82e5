#!/bin/bash
# development helper (GPU box): WRITE_SIZE per env-step and kernel time of the rollout kernel for several batching thresholds
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/ts; cd /tmp; export TMPDIR=/tmp
for th in "$@"; do
  set -- $th
  export BG_TH_PLAY=$1 BG_TH_OTHER=$2 BG_TH_READY=$3
  tag=th_$1_$2_$3
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/ts/$tag --output-format csv -- python3 $R/bench.py --steps 512 --warmup 128 --no-cpu-baseline > $R/gpurun_out/ts/$tag.json 2> $R/gpurun_out/ts/$tag.err
  python3 - <<P
import csv,glob,json
f=glob.glob("$R/gpurun_out/ts/$tag/*/*_counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "rollout2" in r["Kernel_Name"] and r["Counter_Name"]=="WRITE_SIZE"]
v=v[len(v)//4:]
d=json.load(open("$R/gpurun_out/ts/$tag.json"))
print("$tag", "WRITE B/env-step %.0f" % (sum(v)/len(v)*1024/65536/64), "Msteps/s %.0f" % (d["value"]/1e6), "rollout_us %.0f" % d["roofline"]["mean_launch_us"])
P
done
