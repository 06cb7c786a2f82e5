#!/bin/bash
# PC sampling of the engine (beta): where do the waves spend their time?  + the shop-seeding ILP measurement
out=gpurun_out/r05h; mkdir -p $out; export TMPDIR=/tmp
export BALATRO_MI355X_LIB=$PWD/build/variants/lines.so
export T=372 LAUNCHES=3
timeout 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100 --kernel-trace --output-format csv -d $out/pcs_host -o run -- python3 tools/pcs_run.py > $out/pcs_host.log 2>&1; echo "host_trap rc=$?"; tail -3 $out/pcs_host.log
find $out/pcs_host -type f | head; 
python tools/pcs_summary.py $out/pcs_host > $out/pcs_host_summary.txt 2>&1; head -60 $out/pcs_host_summary.txt
timeout 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --kernel-trace --output-format csv -d $out/pcs_sto -o run -- python3 tools/pcs_run.py > $out/pcs_sto.log 2>&1; echo "stochastic rc=$?"; tail -3 $out/pcs_sto.log
python tools/pcs_summary.py $out/pcs_sto > $out/pcs_sto_summary.txt 2>&1; head -40 $out/pcs_sto_summary.txt
# keep the merged output small
for d in pcs_host pcs_sto; do find $out/$d -name "*.csv" -size +20M -delete; done
unset BALATRO_MI355X_LIB
