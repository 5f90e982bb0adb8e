#!/bin/bash
# One GPU measurement cycle (inside gpurun): tools/gpu_cycle.sh <tag> [tests|bench|prof|pmc|train ...]
# Everything lands under gpurun_out/<tag>/ ; copy what is to be kept into profiles/.
TAG=${1:-cycle}; shift
WHAT="${@:-tests bench}"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for w in $WHAT; do
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log ;;
    newtests) timeout 900 python -m pytest tests/test_gpu_bench_size.py tests/test_gpu_parity.py tests/test_gpu_zz_bench_cli.py -m gpu -x -q -s > $OUT/pytest_new.log 2>&1; echo "pytest(new) rc=$?"; tail -5 $OUT/pytest_new.log ;;
    r4tests) timeout 1500 python -m pytest tests/test_gpu_offdist.py tests/test_gpu_trajectory.py tests/test_gpu_streams.py tests/test_gpu_bench_size.py tests/test_gpu_zz_bench_cli.py -m gpu -q -s > $OUT/pytest_r4.log 2>&1; echo "pytest(r4) rc=$?"; grep -E "^\[|passed|failed|FAILED|Error|policy after" $OUT/pytest_r4.log | tail -60 ;;
    f16train) timeout 1500 python -m pytest tests/test_gpu_train_f16.py tests/test_gpu_trajectory.py -m gpu -q -s > $OUT/pytest_f16train.log 2>&1; echo "pytest(f16train) rc=$?"; grep -E "^\[|passed|failed|FAILED|Error|loss scale|unmasked|bf16x3:|f16:" $OUT/pytest_f16train.log | tail -40
           for tp in bf16x3 f16; do timeout 600 python bench.py --train --train-precision $tp --no-cpu-baseline > $OUT/train_$tp.json 2> $OUT/train_$tp.err; echo "train $tp rc=$?"; cat $OUT/train_$tp.json; done ;;
    offdist) timeout 900 python -m pytest tests/test_gpu_offdist.py -m gpu -q -s -k "trained or planted" > $OUT/pytest_offdist.log 2>&1; echo "pytest(offdist) rc=$?"; grep -E "^\[trained|passed|failed|FAILED|policy after|training losses|^E " $OUT/pytest_offdist.log | tail -30 ;;
    varlen) timeout 300 python tools/varlen_bench.py > $OUT/varlen.json 2> $OUT/varlen.err; echo "varlen rc=$?"; cat $OUT/varlen.json | cut -c1-400 ;;
    lat) timeout 300 python tools/latency_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/latency.txt
         timeout 600 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q -k "low_latency" 2>&1 | tail -3 ;;
    bn16) for g in 8192 2048 32768; do DS_TF_GRID=$g timeout 300 python tools/bn16_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bn16_probe.txt; done ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 ;;
    bench) timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json | head -c 6000 ;;
    benchq) timeout 600 python bench.py --no-secondary --no-cpu-baseline > $OUT/benchq.json 2> $OUT/benchq.err; echo "benchq rc=$?"; cat $OUT/benchq.json | head -c 4000 ;;
    train) timeout 600 python bench.py --train --no-cpu-baseline > $OUT/train.json 2> $OUT/train.err; echo "train rc=$?"; cat $OUT/train.json
           MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 timeout 600 python bench.py --train --force-collectives --no-cpu-baseline > $OUT/train_dp.json 2> $OUT/train_dp.err; echo "train_dp rc=$?"; cat $OUT/train_dp.json ;;
    prof) (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -- python $R/bench.py --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline --no-secondary > $OUT/prof.log 2>&1); echo "prof rc=$?"
          python tools/rocpd_stats.py $(find $OUT/prof -name "*.db") > $OUT/kernel_stats.md 2>$OUT/kernel_stats.err; head -40 $OUT/kernel_stats.md
          python tools/step_timeline.py $(find $OUT/prof -name "*.db") > $OUT/step_timeline.txt 2>&1; tail -4 $OUT/step_timeline.txt ;;
    trainprof) (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trainprof -- python $R/bench.py --train --steps 6 --warmup 2 --repeats 0 --no-cpu-baseline > $OUT/trainprof.log 2>&1); echo "trainprof rc=$?"
          python tools/rocpd_stats.py $(find $OUT/trainprof -name "*.db") > $OUT/train_kernel_stats.md 2>$OUT/train_kernel_stats.err; python tools/timeline.py $(find $OUT/trainprof -name "*.db") adagrad | head -12 ;;
    trainprof16) (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trainprof16 -- python $R/bench.py --train --train-precision f16 --steps 6 --warmup 2 --repeats 0 --no-cpu-baseline > $OUT/trainprof16.log 2>&1); echo "trainprof16 rc=$?"
          python tools/rocpd_stats.py $(find $OUT/trainprof16 -name "*.db") > $OUT/train16_kernel_stats.md 2>$OUT/train16_kernel_stats.err; head -45 $OUT/train16_kernel_stats.md; python tools/timeline.py $(find $OUT/trainprof16 -name "*.db") adagrad | head -12 ;;
    dpprof) (cd /tmp && export TMPDIR=/tmp && MASTER_ADDR=127.0.0.1 MASTER_PORT=29572 timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/dpprof -- python $R/bench.py --train --force-collectives --steps 6 --warmup 2 --repeats 0 --no-cpu-baseline > $OUT/dpprof.log 2>&1); echo "dpprof rc=$?"
          python tools/dp_gaps.py $(find $OUT/dpprof -name "*.db") > $OUT/dp_gaps.md 2>$OUT/dp_gaps.err; head -3 $OUT/dp_gaps.md; python tools/timeline.py $(find $OUT/dpprof -name "*.db") adagrad | head -12 ;;
    slots) for c in 4 8 16 32; do timeout 300 python bench.py --no-secondary --no-cpu-baseline --refine-slots $c --repeats 2 > $OUT/slots_$c.json 2> $OUT/slots_$c.err; python -c "import json;d=json.load(open('$OUT/slots_$c.json'));print('slots',$c,d['value'],d['ms_per_step'],d['repeats_ms_per_step'],d['roofline']['achieved'],d['refine'])"; done ;;
    extras) timeout 300 python tools/mine_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mine_probe.txt
            timeout 300 python tools/latency_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/latency.txt ;;
    trainab) for i in 1 2; do for f in "" "--no-fuse-bn"; do echo "train_bench --grouped $f"; timeout 300 python tools/train_bench.py --grouped --steps 20 --warmup 5 $f 2>&1 | grep -v amdgpu.ids | tail -1; done; done | tee $OUT/train_ab.txt ;;
    traintests) timeout 1200 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_parity.py tests/test_gpu_zz_bench_cli.py -m gpu -x -q -s > $OUT/pytest_train.log 2>&1; echo "pytest(train) rc=$?"; tail -15 $OUT/pytest_train.log ;;
    fwdstreams) timeout 300 python tools/train_fwd_streams.py 2>&1 | grep -v amdgpu.ids | tee $OUT/train_fwd_streams.txt ;;
    pmc16) timeout 1500 tools/pmc_run.sh $TAG/pmc16 --train --train-precision f16 --repeats 0; python tools/pmc_train_summary.py $OUT/pmc16 > $OUT/pmc_train16_summary.md 2> $OUT/pmc_train16_summary.err; head -70 $OUT/pmc_train16_summary.md ;;
    pmc) timeout 1200 tools/pmc_run.sh $TAG/pmc --no-secondary --repeats 0; python tools/pmc_summary.py $OUT/pmc kernel 52 > $OUT/pmc_summary.md 2> $OUT/pmc_summary.err; head -60 $OUT/pmc_summary.md ;;
    sideprio) for i in 1 2; do for m in normal low; do DS_SIDE_STREAM_PRIORITY=$m timeout 300 python bench.py --no-secondary --no-cpu-baseline --repeats 3 > $OUT/sideprio_$m.json 2> $OUT/sideprio_$m.err; python -c "import json;d=json.load(open('$OUT/sideprio_$m.json'));print('side stream priority','$m',d['value'],d['ms_per_step'],d['repeats_ms_per_step'],d['roofline']['achieved'],d['roofline']['isolated']['achieved'])"; done; done ;;
    hostprof) timeout 300 python tools/host_profile.py > $OUT/host_profile.txt 2>&1; echo "hostprof rc=$?"; grep -E "host enqueue" $OUT/host_profile.txt
              timeout 300 python tools/host_profile.py --events 2>&1 | grep -E "host enqueue" ;;
    ablayout) timeout 240 python tools/ab_layout.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_layout.txt ;;
    *) echo "unknown step $w" ;;
  esac
done
