#!/bin/bash
# phase clocks of DenseLatticeKernel (a -DRS_DL_PROFILE build in a scratch copy).  usage (GPU box): bash profiles/micro/prof_dense_lattice.sh
cd "$GRAFT_REPO_ROOT"
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
rm -rf /tmp/rsdl && mkdir -p /tmp/rsdl && cp -r rhasspy_speech_amd include /tmp/rsdl/
rm -f /tmp/rsdl/rhasspy_speech_amd/csrc/decode_reg.o
make -C /tmp/rsdl/rhasspy_speech_amd/csrc -j16 EXTRA="-DRS_DL_PROFILE" > /tmp/rsdl/make.log 2>&1 || { tail -20 /tmp/rsdl/make.log; exit 1; }
cp /tmp/rsdl/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
python profiles/micro/nbest_threads.py 2>&1 | grep "dense lattice block" | tail -4
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
