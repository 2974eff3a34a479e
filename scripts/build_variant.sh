# Builds a variant of the HIP library next to the shipped one (same sources, extra flags), for A/B probes and diagnostics:
#   bash scripts/build_variant.sh stamps -DGLS_STAMPS     -> gnngls_amd/libgnngls_hip_stamps.so
# Use it with GNNGLS_HIP_SO=$PWD/gnngls_amd/libgnngls_hip_<name>.so (gnngls_amd/_lib.py).  *.so files are git-ignored.
name=$1; shift
cd "$(dirname "$0")/../gnngls_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result "$@" \
    gls_kernels.hip model_kernels.hip train_kernels.hip capi.hip -o ../libgnngls_hip_$name.so && echo built libgnngls_hip_$name.so
