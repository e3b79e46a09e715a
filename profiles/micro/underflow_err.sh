#!/bin/bash
# profiles/micro/underflow_err.py on a -DRS_TUNING build of the library (RS_GEMM_B3_NOUNDER exists only there)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -r rhasspy_speech_amd include /tmp/rstune/ && cp -r profiles/micro /tmp/rstune/profiles/ && rm -f /tmp/rstune/rhasspy_speech_amd/csrc/*.o
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j16 EXTRA=-DRS_TUNING > /tmp/rstune/make.log 2>&1 || { tail /tmp/rstune/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so; cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
python profiles/micro/underflow_err.py
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
