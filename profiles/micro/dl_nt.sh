#!/bin/bash
# DenseLatticeKernel: workgroup size A/B (-DRS_TUNING build, RS_DL_NT).  usage (GPU box): bash profiles/micro/dl_nt.sh
cd "$GRAFT_REPO_ROOT"
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
rm -rf /tmp/rstune && mkdir -p /tmp/rstune && cp -r rhasspy_speech_amd include profiles /tmp/rstune/
find /tmp/rstune/rhasspy_speech_amd/csrc -name "*.o" -delete
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA="-DRS_TUNING" > /tmp/rstune/make.log 2>&1 || { tail -20 /tmp/rstune/make.log; exit 1; }
cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
for nt in 256 512 1024; do
  echo "RS_DL_NT=$nt"
  RS_DL_NT=$nt RS_LATTICE_TRACE=1 python profiles/micro/nbest_inflight.py 1 4 2>&1 | grep -E "in flight|lattice tail" | awk '/lattice tail/ {k+=$6; n++} /in flight/ {print substr($0,1,48), " mean kernel+count", (n?k/n:0); k=0; n=0}'
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
