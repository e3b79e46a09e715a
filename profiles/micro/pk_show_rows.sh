cd "$GRAFT_REPO_ROOT"
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
rm -rf /tmp/rspk && mkdir -p /tmp/rspk && cp -r rhasspy_speech_amd include /tmp/rspk/
( cd /tmp/rspk/rhasspy_speech_amd/csrc && rm -f feat_kernels.o && make NOPACK="-fno-vectorize" feat_kernels.o && make ) > /tmp/rspk/make.log 2>&1
cp /tmp/rspk/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
STRESS_SHOW_ROWS=1 timeout 600 python profiles/micro/stress_same_model.py 3 4 2>&1 | grep -v "^per\|^mism" | head -40
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
