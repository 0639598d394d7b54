# Kernel-trace evidence (on the GPU box): usage  bash tools/prof_round.sh TAG [gpg]
#   -> gpurun_out/TAG_bench_trace.md (+ TAG_gpg_trace.md); copy into profiles/.
# Every profiler invocation is bounded by its own timeout: a hung rocprofv3 must not eat the GPU budget.
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/prof_kt /tmp/prof_gpg
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-epoch --min-seconds 0.4 > /tmp/kt.log 2>&1; echo "kt rc=$?" )
KT=$(find /tmp/prof_kt -name "*.db" | head -1)
[ -n "$KT" ] && python tools/rocprof_summary.py gpurun_out/${TAG}_bench_trace.md "bench.py kernel trace (infer fp32 + bf16x3 + train legs)=$KT" > /dev/null
# the headline leg alone: in the full command the pipelined config-5 leg runs trunk launches WHILE sampler kernels share the
# chip (their average is then not the kernel's own); this trace is the one roofline.avg_launch_ms must agree with
rm -rf /tmp/prof_kh
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kh -o kh -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-epoch --no-train --no-fast --no-config5 --no-configs > /tmp/kh.log 2>&1; echo "kh rc=$?" )
KH=$(find /tmp/prof_kh -name "*.db" | head -1)
[ -n "$KH" ] && python tools/rocprof_summary.py gpurun_out/${TAG}_bench_trace_headline.md "bench.py --no-train --no-fast --no-config5 --no-configs: the headline inference leg alone=$KH" > /dev/null
python -c "import json; d=[json.loads(l) for l in open('/tmp/kh.log') if l.startswith('{')][-1]; print('bench line of the same run (HIP events on the launch stream): roofline.avg_launch_ms', d['roofline']['avg_launch_ms'], ' ms_per_step', d['ms_per_step'])" >> gpurun_out/${TAG}_bench_trace_headline.md 2>/dev/null
if [ "$2" = "gpg" ]; then
( cd /tmp && BOTH=0 timeout 180 rocprofv3 --kernel-trace --stats -d /tmp/prof_gpg -o gpg -- python $GRAFT_REPO_ROOT/tools/bench_gpg_scale.py > /tmp/gpg.log 2>&1; echo "gpg rc=$?" )
GP=$(find /tmp/prof_gpg -name "*.db" | head -1)
[ -n "$GP" ] && python tools/rocprof_summary.py gpurun_out/${TAG}_gpg_trace.md "tools/bench_gpg_scale.py kernel trace: the GPG sampler on 20,000 sample points of a 50,000-point scene, 6 calls (1 warm-up, 3 timed, 1 staged, the index build)=$GP" > /dev/null
tail -n 3 /tmp/gpg.log
fi
ls -la gpurun_out/ | tail -5
