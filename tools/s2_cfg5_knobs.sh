# cfg5 (Resnet at 32x32 patches, inference): the gathered-product variants behind their A/B switches, one box
O=gpurun_out/s2knobs; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --workload labelprop --model 1 --steps 10 --warmup 3 2> $O/$name.err | grep '^{' > $O/$name.json; echo "$name done"; }
run default A=1
run spec0 CRW_RN_SPEC=0
run spec0_bk64 CRW_RN_SPEC=0 CRW_RN_BK=64
run spec0_bk32 CRW_RN_SPEC=0 CRW_RN_BK=32
run tm256 CRW_RN_TM=256
run spec3 CRW_RN_SPEC=3
run spec4 CRW_RN_SPEC=4
run default2 A=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s2knobs/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'no line', e); continue
    ks = {k['kernel'][:52]: round(k['launch_us'], 1) for k in d.get('roofline_kernels', [])[:5]}
    print(f.split('/')[-1], round(d['ms_per_step'], 3), d['label_agreement_with_oracle'], ks)
PY
