# cfg5 label propagation with the CNN encoder: one step as a kernel listing
set -o pipefail
R=$PWD; O=gpurun_out/s2prof; mkdir -p $O
timeout -k 10 200 python bench.py --workload labelprop --steps 5 --warmup 2 2> $O/lpcnn.err | grep '^{' > $O/lpcnn_line.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lpc && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lpc -o run -- python3 $R/bench.py --workload labelprop --steps 4 --warmup 2 --no-events > $R/$O/prof_lpc.log 2>&1
find /tmp/prof_lpc -name "*kernel_trace.csv" -exec cp {} $R/$O/lpc_trace.csv \;
cd $R && python tools/step_listing.py $O/lpc_trace.csv front_fwd 4 > $O/lpcnn_step_listing.txt 2>&1
rm -f $O/lpc_trace.csv
