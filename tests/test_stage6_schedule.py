"""CPU: the window schedule of stage6.convert_pair(window=...) (the pass-level wavefront inside one utterance pair).
ADVICE r4 (high): with lens (446, 660) at window 224 the decoder pass of window 1 stacked two FINISHING rows of 226 frames beside an
UNFINISHED row of 224 -- the pass ran 226 steps for every row, the unfinished one stepped over latent frames the encoder had not
written and carried the state behind them.  The schedule now never gives an unfinished row fewer frames than the pass runs."""
import numpy as np
import pytest

import stage6


def check(lens, window, reach=4):
    Tmax = max(lens)
    edges = stage6._window_edges(Tmax, window, reach)
    assert edges[0] == 0 and edges[-1] == Tmax and all(b - a > 2 * reach or len(edges) == 2 for a, b in zip(edges, edges[1:]))
    nfr = [lens[0], lens[0], lens[1]]
    sched = stage6._decoder_window_schedule(nfr, edges, reach)
    assert len(sched) == len(edges) - 1
    done = [0, 0, 0]
    for w, (rows, spans) in enumerate(sched):
        stop = edges[w + 1]
        T = max(spans) if spans else 0
        for i, k in zip(rows, spans):
            assert k > 0
            last = done[i] + k
            # a decoder frame t needs latent frames <= t + reach of its utterance: written by encoder windows <= w
            assert last == nfr[i] and nfr[i] <= stop or last + reach <= stop, (lens, window, w, i)
            # frames < T only for a row that finishes here (nothing behind it: the extra steps run over zero padding, state unused)
            assert k == T or last == nfr[i], (lens, window, w, i, spans)
            done[i] = last
    assert done == nfr


def test_the_reported_case_and_its_neighbours():
    assert stage6._decoder_window_schedule([446, 446, 660], [0, 224, 448, 660], 4)[1] == ([0, 1, 2], [224, 224, 224])
    for a in range(440, 452):
        check((a, 660), 224)
        check((660, a), 224)


def test_random_length_pairs_and_windows():
    rng = np.random.RandomState(5)
    for _ in range(3000):
        lens = (int(rng.randint(1, 1600)), int(rng.randint(1, 1600)))
        w = int(rng.randint(9, 400))
        check(lens, w if rng.rand() < 0.7 else [int(rng.randint(9, 300)), w])


def test_windows_must_exceed_twice_the_reach():
    with pytest.raises(ValueError):
        stage6._window_edges(100, 8, 4)
