cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O2 -DPK_LIBRARY -shared -fPIC -o /tmp/libpkvictim.so profiles/micro/pk_repro2.hip 2>/dev/null
for e in "A=1" "RS_GEMM_B3=0" "RS_GEMM_B3J=0" "RS_GEMM_B3J=0 RS_GEMM_B3I=0" "RS_IVEC_MFMA=0 RS_UBM_MFMA=0" "RS_DECODER=sparse" "RS_DECODER=dense"; do
  echo "== $e: $(env $e timeout 300 python profiles/micro/pk_victim_beside_decode.py 2>&1 | grep 'beside three' | head -1 | cut -c1-150)"
done
