# Evidence of a round (GPU box): full GPU suite, smoke, default bench, per-config / scoring / sampler / loader benches,
# kernel traces and PMC passes.  usage: bash tools/final_round.sh TAG     Every profiler call is bounded by a timeout.
TAG=${1:-r06}
FAILED=""
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -2 | tee gpurun_out/${TAG}_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_gpu_suite.txt
python bench.py > gpurun_out/${TAG}_bench_final.json 2> /tmp/bench.err      # traffic measured in-run by default; echo "bench rc=$?"; cut -c1-300 gpurun_out/${TAG}_bench_final.json
timeout 300 python tools/bench_configs.py > gpurun_out/${TAG}_bench_configs.jsonl 2> /tmp/cfg.err; echo "cfg rc=$?"
timeout 120 python tools/bench_latency.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_latency.json
timeout 200 python tools/bench_gpg.py --P 3000 20000 50000 --cpu-draws 2 2>/dev/null | tail -3 > gpurun_out/${TAG}_bench_gpg.jsonl
timeout 200 python tools/bench_scoring.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_scoring.json
timeout 200 python tools/bench_pipeline.py 2>/dev/null | tail -3 > gpurun_out/${TAG}_bench_pipeline.jsonl
timeout 120 python tools/bench_loader.py 2>/dev/null > gpurun_out/${TAG}_bench_loader.jsonl
timeout 200 python tools/bench_epoch.py 2>/dev/null > gpurun_out/${TAG}_bench_epoch.jsonl
timeout 200 python tools/bench_gpg_scale.py 2>/dev/null > gpurun_out/${TAG}_bench_gpg_scale.jsonl
timeout 100 python tools/bench_crop.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_crop.json
FUSED_LOSS=1 timeout 100 python tools/find_copies.py 2>/dev/null > gpurun_out/${TAG}_aten_launches_in_a_step.txt; echo "(empty = no ATen launch inside a training step)" >> gpurun_out/${TAG}_aten_launches_in_a_step.txt
timeout 600 python tools/localise_residual.py > gpurun_out/${TAG}_localise_residual.txt 2>/dev/null
timeout 100 python tools/bench_pass.py 2>/dev/null > gpurun_out/${TAG}_bench_pass.txt
bash tools/prof_round.sh ${TAG} gpg > /tmp/prof.log 2>&1; grep "rc=" /tmp/prof.log
bash tools/trace_train.sh ${TAG} > /tmp/tr.log 2>&1; grep "rc=" /tmp/tr.log
bash tools/trace_train.sh ${TAG} bf16 > /tmp/tr.log 2>&1; grep "rc=" /tmp/tr.log
bash tools/trace_train.sh ${TAG} bf16x3 > /tmp/tr.log 2>&1; grep "rc=" /tmp/tr.log
bash tools/pmc_round.sh ${TAG} hbm > /tmp/pmc.log 2>&1; grep "rc=" /tmp/pmc.log
bash tools/pmc_train.sh ${TAG} > /tmp/pmct.log 2>&1; grep "rc=" /tmp/pmct.log
# the strong-scaling table: stderr is KEPT (RCCL's banner goes there; r05 swallowed a failure and committed an empty table)
timeout 400 python tools/bench_strong.py > gpurun_out/${TAG}_bench_strong.jsonl 2> gpurun_out/${TAG}_bench_strong.err; rc=$?
echo "bench_strong rc=$rc rows=$(grep -c step_ms gpurun_out/${TAG}_bench_strong.jsonl)"
if [ $rc -ne 0 ] || ! grep -q projection gpurun_out/${TAG}_bench_strong.jsonl; then echo "FAILED: bench_strong (see gpurun_out/${TAG}_bench_strong.err)"; tail -5 gpurun_out/${TAG}_bench_strong.err; FAILED="$FAILED bench_strong"; fi
timeout 120 ./examples/cabi_index_consumer > gpurun_out/${TAG}_index_consumer.txt 2>&1
# GPU AddressSanitizer (xnack+ code objects, HSA_XNACK=1) is refused on this pool since round 6: profiles/r05_asan.txt is the last run
PNGPD_GATE_DIAG=1 timeout 1200 python -m pytest tests/test_gpu_grad_gate.py -m gpu -q -s 2>&1 | grep "gate B=\|passed\|failed" | cut -c1-6000 > gpurun_out/${TAG}_gate_diag.txt
timeout 600 python -m pytest tests/test_gpu_head_train.py tests/test_gpu_refine.py tests/test_gpu_rccl.py -m gpu -q -s 2>&1 | grep "^\[\|head B=\|refine\|passed\|failed" | cut -c1-400 > gpurun_out/${TAG}_gates_head_refine_rccl.txt
for B in 128; do rm -rf /tmp/pt$B; ( cd /tmp && TRACE_B=$B timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pt$B -o t -- python $GRAFT_REPO_ROOT/tools/trace_train.py 20 fp32 > /tmp/tt$B.log 2>&1 ); DB=$(find /tmp/pt$B -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py --all gpurun_out/${TAG}_train_step_trace_B$B.md "20 eager training steps at B $B N 1024 fp32 (tools/trace_train.py)=$DB" > /dev/null; done
timeout 120 python tools/bench_eval.py 2>/dev/null > gpurun_out/${TAG}_bench_eval.txt
timeout 120 python tools/bench_step.py 2>/dev/null > gpurun_out/${TAG}_bench_step.txt
for f in trace_small_train trace_small_eval; do rm -rf /tmp/pst; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pst -o t -- python $GRAFT_REPO_ROOT/tools/$f.py > /tmp/$f.log 2>&1 ); DB=$(find /tmp/pst -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py --all gpurun_out/${TAG}_${f}.md "$f (B 64 N 750)=$DB" > /dev/null; done
# matrix / vector issue probe (tools/probes/mfma_valu_overlap.hip, built on the dev box into build_probe/)
( for v in 0 1 4 8 9; do [ -x build_probe/mvo$v ] && timeout 60 ./build_probe/mvo$v; done ) > gpurun_out/${TAG}_probe_mfma_valu.txt 2>&1
# per-phase wave-cycle accounting of passes C / D / E (a -DPNGPD_TIMING build of the library in build_probe/)
[ -f build_probe/lib_tm.so ] && PNGPD_LIB=$GRAFT_REPO_ROOT/build_probe/lib_tm.so timeout 200 python tools/phase_times.py 2>/dev/null | grep -v "^B " > gpurun_out/${TAG}_phase_times.txt
# round 6: the bf16-mode passes — per-pass times + rooflines, A/B against the round-5 library when one was shipped
# (build_probe/lib_base.so: the library at the commit before pngpd_bwd_bf.h / pair conversion), phase stamps, tr probe
timeout 200 python tools/bench_pass_bf.py 2>/dev/null > gpurun_out/${TAG}_bench_pass_bf.jsonl
if [ -f build_probe/lib_base.so ]; then
  ( for i in 1 2; do
      echo "== round-5 kernels (build_probe/lib_base.so), run $i"; PNGPD_LIB=$GRAFT_REPO_ROOT/build_probe/lib_base.so timeout 200 python tools/bench_pass_bf.py 2>/dev/null | python -c "import sys,json; [print(d['nterms'], d['ms']) for d in map(json.loads, sys.stdin)]"
      echo "== this tree, run $i"; timeout 200 python tools/bench_pass_bf.py 2>/dev/null | python -c "import sys,json; [print(d['nterms'], d['ms']) for d in map(json.loads, sys.stdin)]"
    done ) > gpurun_out/${TAG}_ab_passes.txt 2>&1
fi
[ -f build_probe/lib_tm.so ] && PNGPD_LIB=$GRAFT_REPO_ROOT/build_probe/lib_tm.so timeout 200 python tools/phase_times_bf.py 2>/dev/null > gpurun_out/${TAG}_phase_times_bf.txt
[ -f build_probe/lib_tm.so ] && PNGPD_LIB=$GRAFT_REPO_ROOT/build_probe/lib_tm.so timeout 200 python tools/phase_times_c_x3.py 2>/dev/null > gpurun_out/${TAG}_phase_times_c_x3.txt
[ -x build_probe/ds_tr_probe ] && ./build_probe/ds_tr_probe 16 0 > gpurun_out/${TAG}_ds_tr_probe.txt 2>&1
# round 6: in-box counts / crop kernel times on the dense (sampled-candidates) scene; distinct arg-max points per cloud
timeout 200 python tools/bench_small_graph.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_small_graph.json
timeout 200 python tools/probe_crop_counts.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_probe_crop_counts.json
timeout 200 python tools/probe_unique_args.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_probe_unique_args.json
ls gpurun_out | grep ${TAG}
# every evidence file must be non-trivial: an empty / banner-only file is reported, never silently committed
for f in gpurun_out/${TAG}_*; do case $f in *_aten_launches_in_a_step.txt) continue;; esac; [ $(wc -c < $f) -lt 64 ] && { echo "SUSPICIOUS (under 64 bytes): $f"; FAILED="$FAILED $f"; }; done
[ -n "$FAILED" ] && { echo "final_round: FAILED STEPS:$FAILED"; exit 1; }
echo "final_round: all steps produced evidence"
