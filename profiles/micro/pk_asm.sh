# ISA-level experiments on the SLP-vectorized MfccKernel: the device assembly of feat_kernels.hip is edited (pk_asm_patch.py),
# assembled, linked and bundled by hand, the host side compiled against it, the library relinked in a scratch copy, the
# concurrent-call stress test run.   usage (GPU box): bash profiles/micro/pk_asm.sh <out> <iterations> <variant>...
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pk_asm}; IT=${2:-30}
shift; shift
mkdir -p $OUT
L=/opt/rocm/lib/llvm/bin
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
for v in "$@"; do
  rm -rf /tmp/rspk && mkdir -p /tmp/rspk && cp -r rhasspy_speech_amd include /tmp/rspk/
  ( cd /tmp/rspk/rhasspy_speech_amd/csrc
    FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result -ffp-contract=off -fno-vectorize"
    hipcc $FL -g1 --offload-device-only -S -o dev.s feat_kernels.hip
    python "$GRAFT_REPO_ROOT"/profiles/micro/pk_asm_patch.py dev.s dev2.s $v feat_kernels.hip
    $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c dev2.s -o dev.o
    $L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared dev.o -o dev.out
    $L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=dev.out -output=dev.hipfb
    hipcc $FL --offload-host-only -Xclang -fcuda-include-gpubinary -Xclang dev.hipfb -c feat_kernels.hip -o feat_kernels.o
    touch feat_kernels.o && make ) > /tmp/rspk/make.log 2>&1
  grep "packed operations" /tmp/rspk/make.log | tail -1
  cp /tmp/rspk/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  echo "SLP build, assembly variant $v: $(timeout 600 python profiles/micro/stress_same_model.py $IT 4 2>&1 | tail -2 | cut -c1-60,170-260 | tr '\n' ' ')" | tee -a $OUT/result.txt
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
