"""Randomised decode cases (tests/test_gpu_fuzz_decode.py, oracle/gen_fuzz_decode_golden.py): the case dictionaries of
tests/cases.py drawn from a seed -- model shape, graph kind, utterance, decoder options.  Deterministic: the same list here and
in the build container where the reference produced tests/golden/fuzz_decode.json."""
import numpy as np

N_CASES = 96


def _draw(i: int) -> dict:
    rng = np.random.default_rng(7000 + i)
    spec = dict(seed=int(rng.integers(1, 1000)), num_phones=int(rng.integers(24, 41)))
    if rng.random() < 0.25:
        spec["ivector_dim"] = 0
    elif rng.random() < 0.3:
        spec["nnet_cmvn"] = True
    if rng.random() < 0.3:
        spec.update(tdnnf=True, layer_offsets=((0,), (-1, 0, 1), (-1, 0, 1), (-3, 0, 3)))
    else:
        spec["layer_offsets"] = [((0,), (-1, 0, 1), (-2, 0, 2)), ((0,), (-1, 0, 1), (-1, 0, 1), (-3, 0, 3)), ((0,), (-2, 0, 2))][int(rng.integers(0, 3))]
    if rng.random() < 0.3:
        spec.update(with_priors=True, with_log_softmax=True)
    if rng.random() < 0.3:
        spec["chain_topology"] = False
    if rng.random() < 0.2:
        spec["binary"] = False
    spec["hidden_dim"] = int(rng.choice([24, 32, 48]))
    r = rng.random()
    if r < 0.55:
        graph = "grammar" if rng.random() < 0.8 else "grammar:vector"
    else:
        graph = f"arpa:{int(rng.integers(20, 200))}:{int(rng.integers(50, 800))}"
    case = dict(spec=spec, graph=graph, audio=f"synth:{int(rng.integers(100, 100000))}:{int(rng.integers(6000, 64000))}")
    if rng.random() < 0.5:
        opts = dict(beam=float(rng.choice([8.0, 12.0, 16.0, 24.0])), lattice_beam=float(rng.choice([4.0, 6.0, 8.0])))
        if rng.random() < 0.5:
            opts.update(max_active=int(rng.choice([40, 100, 400, 7000])), min_active=int(rng.choice([0, 20, 200])))
        case["opts"] = opts
    return case


CASES = [_draw(i) for i in range(N_CASES)]
