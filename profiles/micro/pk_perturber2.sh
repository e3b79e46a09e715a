# What in GemmKernelB3 does the victim (pk_repro2.hip: v_pk_mul_f32 with source 1 read half-swapped) need beside it?  Scratch builds of
# the library with one ingredient of the kernel compiled out (-DRS_B3_ABLATE: results wrong by design, the decode threads only have to
# keep the kernels running), every wide layer on GemmKernelB3 (RS_GEMM_B3J=0 RS_GEMM_B3I=0).   usage (GPU box): bash profiles/micro/pk_perturber2.sh [bits...]
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O2 -DPK_LIBRARY -shared -fPIC -o /tmp/libpkvictim.so profiles/micro/pk_repro2.hip 2>/dev/null
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
for b in ${@:-0 1 2 4 8 16}; do
  rm -rf /tmp/rspk && mkdir -p /tmp/rspk && cp -r rhasspy_speech_amd include /tmp/rspk/
  ( cd /tmp/rspk/rhasspy_speech_amd/csrc && rm -f nnet_gemm_b3.o && make EXTRA=-DRS_B3_ABLATE=$b nnet_gemm_b3.o && make EXTRA= ) > /tmp/rspk/make.log 2>&1
  cp /tmp/rspk/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  echo "RS_B3_ABLATE=$b: $(RS_GEMM_B3J=0 RS_GEMM_B3I=0 timeout 300 python profiles/micro/pk_victim_beside_decode.py 2>&1 | grep 'beside three' | head -1 | cut -c30-140)"
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
