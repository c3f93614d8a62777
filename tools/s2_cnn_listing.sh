# one CNN training step (bench.py default) as a kernel listing with the idle gaps (tools/step_listing.py)
set -o pipefail
R=$PWD; O=gpurun_out/s2prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cnn && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cnn -o run -- python3 $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-probe > $R/$O/prof_cnn.log 2>&1
find /tmp/prof_cnn -name "*kernel_trace.csv" -exec cp {} $R/$O/cnn_trace.csv \;
cd $R && python tools/step_listing.py $O/cnn_trace.csv front_fwd_kernel 3 > $O/cnn_step_listing.txt 2>&1
rm -f $O/cnn_trace.csv
