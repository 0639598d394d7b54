# A/B timing of library variants built into pointnetgpd_amd/csrc/build/variants/ (scratch experiments, not product):
#   bash tools/abl_run.sh "fwd_train,bwd_d,bwd_e" v1 v2 ...
cd $GRAFT_REPO_ROOT
PASSES=$1; shift
for v in "$@"; do
  echo "== $v"; PNGPD_LIB=$GRAFT_REPO_ROOT/pointnetgpd_amd/csrc/build/variants/lib_$v.so PNGPD_PASSES=$PASSES timeout 100 python tools/bench_pass.py 2>/dev/null | grep -v "^B "
done
