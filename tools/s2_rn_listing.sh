# one Resnet training step (bench.py --model 1) as a kernel listing with the idle gaps (tools/step_listing.py)
set -o pipefail
R=$PWD; O=gpurun_out/s2prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_rn && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rn -o run -- python3 $R/bench.py --model 1 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-probe > $R/$O/prof_rn.log 2>&1
find /tmp/prof_rn -name "*kernel_trace.csv" -exec cp {} $R/$O/rn_trace.csv \;
cd $R && python tools/step_listing.py $O/rn_trace.csv rn_pack_stem_frag 3 > $O/rn_step_listing.txt 2>&1
rm -f $O/rn_trace.csv
