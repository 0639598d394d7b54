# libpngpd under AddressSanitizer on the GPU box (SURVEY.md §5's sanitizer target).
#   build (container, cross-compiles):  make -C pointnetgpd_amd/csrc asan     -> csrc/build/asan/libpngpd_asan.so
#   (host code AND the gfx950 kernels instrumented; kernels built for gfx950:xnack+, run with HSA_XNACK=1)
#   the build directory must travel: remove "pointnetgpd_amd/csrc/build/asan/" from .gpurunignore for this call.
# What runs: the torch-free C-ABI consumers linked against the ASan library — examples/cabi_consumer.cpp (the inference
# trunk + FC layer, and with "train" one training step through pngpd_trunk_train_fwd/_bwd + pngpd_adam_flat) and
# examples/cabi_index_consumer.cpp (the index-heavy crop / GPG / GPD kernels on edge-case shapes).
# The python GPU tests cannot run under it in this image: torch bundles its own (uninstrumented) libamdhip64 and the
# image has no ASan-instrumented ROCm runtime (/opt/rocm/lib/asan), so the ASan runtime's hsa_amd_memory_pool_allocate
# interceptor aborts at torch's first device allocation (recorded below).
#   usage: bash tools/asan_run.sh TAG   -> gpurun_out/TAG_asan.txt
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
A=$GRAFT_REPO_ROOT/pointnetgpd_amd/csrc/build/asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
{
  echo "# libpngpd under AddressSanitizer ($(cat .commit_id 2>/dev/null)): $A/libpngpd_asan.so, runtime $RT, HSA_XNACK=1"
  ls -la $A/libpngpd_asan.so $A/cabi_consumer_asan $A/cabi_index_consumer_asan
  for args in "5 200" "16 750 train" "3 64 train" "70 130 train"; do
    echo "== cabi_consumer_asan $args   (ASAN_OPTIONS=detect_leaks=0)"
    LD_LIBRARY_PATH=$(dirname $RT):/opt/rocm/lib HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0 timeout 200 $A/cabi_consumer_asan $args 2>&1 | grep -v "^pool\|^fc " | tail -22
    echo "exit code: ${PIPESTATUS[0]}"
  done
  echo "== cabi_index_consumer_asan   (crop count/compact/ranges/gather/resample + indexed / one-launch crop, batch_keep_rows / stack_gather_lists / train_batch, GPG moments (+indexed) / enumerate / sweep / select / pushin / finish + the fused sweep_select / pushin_sweep, GPD projection, depth registration, conv5 stem + its training form and backward, split-K fc1)"
  LD_LIBRARY_PATH=$(dirname $RT):/opt/rocm/lib HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0 timeout 400 $A/cabi_index_consumer_asan 2>&1 | tail -30
  echo "exit code: ${PIPESTATUS[0]}"
  echo "== python (torch) with the ASan runtime preloaded: torch's bundled HIP runtime is not instrumented"
  HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$RT PNGPD_LIB=$A/libpngpd_asan.so timeout 120 python -c "import torch; torch.zeros(4, device='cuda')" 2>&1 | grep -v amdgpu.ids | head -6
} > gpurun_out/${TAG}_asan.txt 2>&1
cat gpurun_out/${TAG}_asan.txt
