# cfg5 label propagation with the Resnet encoder under rocprofv3: kernel summary, GPU-busy timeline and one step as a listing
set -o pipefail
R=$PWD; O=gpurun_out/s2prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_lp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lp -o run -- python3 $R/bench.py --workload labelprop --model 1 --steps 5 --warmup 2 --no-events > $R/$O/prof_lp.log 2>&1
find /tmp/prof_lp -name "*kernel_stats.csv" -exec cp {} $R/$O/labelprop_resnet_kernel_stats.csv \;
find /tmp/prof_lp -name "*kernel_trace.csv" -exec cp {} $R/$O/lp_trace.csv \;
cd $R && python tools/step_listing.py $O/lp_trace.csv rn_stem_moments_kernel 3 > $O/lp_step_listing.txt 2>&1
rm -f $O/lp_trace.csv
