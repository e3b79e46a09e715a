"""Golden vectors for rhasspy_speech_amd.meta.decode_meta / encode_meta: random transcripts with meta words run through the
reference's own functions (rhasspy_speech/hassil_fst.py:849-876, imported from /root/reference in the build container).
Writes tests/golden/meta_vectors.json.  Test infrastructure: nothing in the product path reads it."""
import json
import random
import sys
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def reference_functions():
    # the package imports hassil / unicode_rbnf at the top (absent here); none of the three functions needs them
    for name in ["hassil", "hassil.expression", "hassil.intents", "hassil.util", "hassil.recognize", "unicode_rbnf"]:
        m = types.ModuleType(name)

        class _Any:
            def __init__(self, *a, **k):
                pass

            def __getattr__(self, k):
                return _Any()

            def __call__(self, *a, **k):
                return _Any()

        m.__getattr__ = lambda k, _A=_Any: _A
        sys.modules[name] = m
    sys.path.insert(0, "/root/reference")
    from rhasspy_speech import hassil_fst as ref
    return ref.decode_meta, ref.encode_meta, ref.decode_meta_single, ref.OUTPUT_PREFIX, ref.SENTENCE_OUTPUT


def main():
    decode_meta, encode_meta, decode_single, OUT, SENT = reference_functions()
    rng = random.Random(4)
    words = ["turn", "on", "the", "kitchen", "light", "set", "to", "50", "%", "überall", "naïve", "{x}", "__output:", "__sentence_output:", "a=b"]
    lists = [None, "name", "area", "brightness", ""]
    vectors = []
    for i in range(200):
        parts = []
        for _ in range(rng.randint(0, 7)):
            r = rng.random()
            if r < 0.35:
                rec = {"text": " ".join(rng.sample(words[:11], rng.randint(0, 3))), "list": rng.choice(lists)}
                parts.append(encode_meta(json.dumps(rec), OUT))
            elif r < 0.45 and not any(p.startswith(SENT) for p in parts):
                tmpl = rng.choice(["{name} in {area}", "plain text", "{brightness}%", "{area}", ""])
                parts.append(encode_meta(tmpl, SENT))
            else:
                parts.append(rng.choice(words))
        text = " ".join(parts)
        try:
            vectors.append({"input": text, "output": decode_meta(text)})
        except Exception as e:  # a template naming a slot the transcript did not fill: the reference raises
            vectors.append({"input": text, "raises": type(e).__name__})
    enc = [{"input": w, "output": encode_meta(w)} for w in words[:12]]
    (ROOT / "tests" / "golden" / "meta_vectors.json").write_text(json.dumps({"decode_meta": vectors, "encode_meta": enc}, indent=0, ensure_ascii=False) + "\n")
    print(len(vectors), "vectors,", sum("raises" in v for v in vectors), "raising")


if __name__ == "__main__":
    main()
