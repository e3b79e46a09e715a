#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY (never linked/loaded by the product library).
#
# Compiles the reference's own CPU Kaldi hot-path binaries straight from the
# sources where they lie under /root/reference (nothing is copied into this
# repo) into oracle/_ref/.  Not the reference's build system: a flat g++
# recipe.  BLAS/LAPACK = the LP64 OpenBLAS that ships inside SciPy in this
# image (real library, symbols prefixed scipy_); a generated rename header maps
# cblas_*/LAPACK names to it.
#
# Outputs (git-ignored, travel to the GPU box with gpurun):
#   oracle/_ref/libkaldi_ref.so       all Kaldi libs on the path + OpenFst
#   oracle/_ref/bin/<tool>            reference mains + our dump drivers
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${RS_REFERENCE_DIR:-/root/reference}"
OUT="$HERE/_ref"
K="$REF/kaldi/src"
F="$REF/kaldi/openfst/src"
if [ ! -d "$K" ]; then echo "reference tree not present ($K): skipping oracle/_ref build"; exit 0; fi
LIBDIR="$(python3 -c 'import scipy,os;print(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)),"scipy.libs"))')"
BLAS="$(ls "$LIBDIR"/libscipy_openblas*.so | head -1)"
JOBS="${JOBS:-8}"
mkdir -p "$OUT/obj" "$OUT/bin"

# 1. rename header cblas_x -> scipy_cblas_x (+ LAPACK routines Kaldi uses)
nm -D "$BLAS" | awk '$2=="T" && $3 ~ /^scipy_(cblas_|[sd](getrf|getri|sptrf|sptri|tptrf|tptri|gesvd|trtri)_$)/ {s=$3; sub(/^scipy_/,"",s); print "#define " s " " $3}' > "$OUT/blas_rename.h"

CXXFLAGS=(-std=c++14 -O2 -fPIC -w -DHAVE_CLAPACK=1 -DKALDI_NO_PORTAUDIO=1 '-DKALDI_VERSION="oracle"'
          -include "$OUT/blas_rename.h" -I"$K" -I"$F/include" -I"$REF/kaldi/tools/CLAPACK")

compile_one() {  # src -> obj (skips up-to-date)
  local src="$1" obj="$OUT/obj/$(echo "$1" | sed "s#^$REF/##; s#/#__#g; s#\.cc\$#.o#")"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then g++ "${CXXFLAGS[@]}" -c "$src" -o "$obj"; fi
}
export -f compile_one; export OUT REF K F
export CXXFLAGS_STR="$(printf '%q ' "${CXXFLAGS[@]}")"

list=()
for f in "$F"/lib/*.cc; do list+=("$f"); done
for d in base matrix util feat tree gmm transform fstext hmm lm decoder lat cudamatrix nnet3 chain ivector online2; do
  for f in "$K/$d"/*.cc; do
    case "$f" in *-test.cc|*test-utils*|*/online-nnet2-decoding*.cc) continue;; esac  # nnet2 decoders are not on the path
    list+=("$f")
  done
done
printf '%s\n' "${list[@]}" | xargs -P "$JOBS" -I{} bash -c '
  src="{}"; obj="$OUT/obj/$(echo "$src" | sed "s#^$REF/##; s#/#__#g; s#\.cc\$#.o#")"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then eval g++ $CXXFLAGS_STR -c "$src" -o "$obj"; fi'

g++ -shared -o "$OUT/libkaldi_ref.so" "$OUT"/obj/*.o -L"$LIBDIR" -l:"$(basename "$BLAS")" -Wl,-rpath,"$LIBDIR" -lpthread -ldl

link_tool() {  # src name
  g++ "${CXXFLAGS[@]}" "$1" -o "$OUT/bin/$2" -L"$OUT" -lkaldi_ref -Wl,-rpath,'$ORIGIN/..' \
      -L"$LIBDIR" -l:"$(basename "$BLAS")" -Wl,-rpath,"$LIBDIR" -lpthread -ldl
}
for t in online2bin/online2-wav-nnet3-latgen-faster online2bin/online2-cli-nnet3-decode-faster \
         latbin/lattice-to-nbest latbin/nbest-to-linear latbin/lattice-copy \
         gmmbin/gmm-init-mono nnet3bin/nnet3-init nnet3bin/nnet3-am-init nnet3bin/nnet3-am-info \
         nnet3bin/nnet3-am-copy bin/show-transitions bin/copy-matrix featbin/compute-mfcc-feats \
         gmmbin/gmm-global-copy ivectorbin/ivector-extractor-copy \
         latbin/lattice-scale latbin/lattice-to-phone-lattice latbin/lattice-compose latbin/lattice-determinize \
         latbin/lattice-add-trans-probs latbin/lattice-best-path fstbin/fstdeterminizestar fstbin/fstrmsymbols \
         fstbin/fsttablecompose fstbin/fstminimizeencoded fstbin/fstpushspecial fstbin/fstcomposecontext fstbin/fstrmepslocal \
         fstbin/fstaddselfloops fstbin/fstisstochastic bin/make-h-transducer bin/add-self-loops bin/tree-info bin/am-info; do
  [ -f "$K/$t.cc" ] || { echo "missing $t"; continue; }
  [ -x "$OUT/bin/$(basename "$t")" ] && [ "$OUT/bin/$(basename "$t")" -nt "$OUT/libkaldi_ref.so" ] && continue      # already linked
  link_tool "$K/$t.cc" "$(basename "$t")" &
  while [ "$(jobs -r | wc -l)" -ge "$JOBS" ]; do sleep 0.2; done
done
wait
# our own dump drivers (sources live in oracle/drivers, compiled against the reference headers)
for d in "$HERE"/drivers/*.cc; do
  [ -f "$d" ] || continue
  link_tool "$d" "$(basename "${d%.cc}")"
done
# 4. the OpenFst command-line tools the reference's fuzzy matcher pipes an n-best list through (transcribe_util.py:47-60,
#    kaldi.py:391-408): libfstscript + eight mains, from the vendored OpenFst sources
if [ "${RS_BUILD_FST_TOOLS:-1}" = "1" ] && [ -d "$F/script" ]; then
  FSTFLAGS=(-std=c++14 -O2 -fPIC -w -I"$F/include")
  export FSTFLAGS_STR="$(printf '%q ' "${FSTFLAGS[@]}")"
  mkdir -p "$OUT/obj_fst"
  ls "$F"/script/*.cc | xargs -P "$JOBS" -I{} bash -c '
    src="{}"; obj="$OUT/obj_fst/$(basename "${src%.cc}").o"
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then eval g++ $FSTFLAGS_STR -c "$src" -o "$obj"; fi'
  g++ -shared -o "$OUT/libfstscript_ref.so" "$OUT"/obj_fst/*.o
  for t in fstcompile fstarcsort fstcompose fstshortestpath fstrmepsilon fsttopsort fstproject fstprint fstconvert fstequivalent fstinfo \
           fstdeterminize fstminimize; do      # (the last two: KaldiTrainer._create_grammar, kaldi.py:321-341)
    if [ ! -f "$OUT/bin/$t" ] || [ "$F/bin/$t.cc" -nt "$OUT/bin/$t" ]; then
      g++ "${FSTFLAGS[@]}" "$F/bin/$t.cc" "$F/bin/$t-main.cc" -o "$OUT/bin/$t" -L"$OUT" -lfstscript_ref -lkaldi_ref -Wl,-rpath,'$ORIGIN/..' \
          -L"$LIBDIR" -l:"$(basename "$BLAS")" -Wl,-rpath,"$LIBDIR" -lpthread -ldl &
      while [ "$(jobs -r | wc -l)" -ge "$JOBS" ]; do sleep 0.2; done
    fi
  done
  wait
fi
echo "oracle/_ref built: $(ls "$OUT/bin" | wc -l) tools"
