"""CPU: the host-side decisions of stage4.Stage4Step(sync=False) under data parallelism, on a simulated device.
ADVICE r4 (medium): ranks notice a raised status word at different host times and clear their latches a few steps apart; the
in-place MAX all-reduce then re-raised the latch of the rank that had cleared first, which read the same code as a SECOND incident
(double-counted fallbacks; for a time-out code it raised on that rank alone and left the others hanging in the next collective).
Now the reduced word is a separate copy and an incident is a maximal RUN of raised steps, handled once at its first step -- the
same step on every rank."""
import pytest

import stage4

torch = pytest.importorskip("torch")


class FakeEvent(object):
    def __init__(self, sim, rank, step):
        self.sim, self.rank, self.step = sim, rank, step

    def query(self):            # the host of `rank` sees a step `lag` steps late
        return self.step <= self.sim.now - self.sim.lag[self.rank]

    def synchronize(self):
        self.sim.now = max(self.sim.now, self.step + self.sim.lag[self.rank])


class FakeLatch(object):
    """The local device latch of one rank: zero_() is stream-ordered, i.e. it acts before the next step this rank enqueues."""

    def __init__(self):
        self.value, self.clear_pending = 0, False

    def zero_(self):
        self.clear_pending = True


class Rank(stage4.Stage4Step):
    def __init__(self, sim, rank):      # (no modules, no device: only what _lagged_check / _drain touch)
        self.sim, self.rank = sim, rank
        self.fused, self._pending, self._fp32_left, self._incident, self._owns_status = True, [], 0, 0, False
        self.skipped = self.fallbacks = 0
        self.coop_fallback = False
        self.status_dev = FakeLatch()
        self.fp32_switches = []
        self._incident_left, self.verbose_incidents = 0, False
        self.data_parallel = len(sim.lag) > 1

    def _collective(self):
        return self.data_parallel

    def _fp32_reverse(self, on):
        self.fp32_switches.append((self.sim.now, on))

    def _enable_coop_launch(self):
        self.coop_fallback = True

    def _own_status(self):
        self._owns_status = True

    def _release_status(self):
        self._owns_status = False


class Sim(object):
    def __init__(self, lag, raises):
        """lag[r]: steps by which rank r's host trails its device; raises: {(step, rank): code} local kernel reports."""
        self.lag, self.raises, self.now = lag, raises, 0
        self.ranks = [Rank(self, r) for r in range(len(lag))]
        self.reduced = []

    def run(self, steps):
        for k in range(steps):
            self.now = k
            for r in self.ranks:
                r._lagged_check()
            for r in self.ranks:                  # the device side of step k: clear (if enqueued), latch, reduce, gate, slot
                L = r.status_dev
                if L.clear_pending:
                    L.value, L.clear_pending = 0, False
                L.value = max(L.value, self.raises.get((k, r.rank), 0))
            red = max(r.status_dev.value for r in self.ranks)
            self.reduced.append(red)
            for r in self.ranks:
                slot = torch.tensor([red, 0, 0, 0], dtype=torch.int32)
                r._own_status()
                r._pending.append((FakeEvent(self, r.rank, k), slot))
        for r in self.ranks:
            r.finish()


def test_overflow_on_one_rank_is_one_incident_on_every_rank():
    sim = Sim(lag=[1, 4], raises={(10, 1): 5})
    sim.run(60)
    a, b = sim.ranks
    assert a.fallbacks == b.fallbacks == 1
    assert a.skipped == b.skipped == sum(1 for v in sim.reduced if v)          # every rank counts exactly the steps the device skipped
    assert [on for _, on in a.fp32_switches][0] is True and [on for _, on in b.fp32_switches][0] is True
    assert sim.reduced[10] == 5 and sim.reduced[-1] == 0 and not a._owns_status and not b._owns_status
    # the run of raised steps ends once the slower rank has cleared: lag 4 -> noticed when step 14 is enqueued, cleared in front of it
    assert sim.reduced[10:16] == [5, 5, 5, 5, 0, 0]


def test_time_out_is_handled_once_and_nobody_raises_alone():
    sim = Sim(lag=[0, 5], raises={(7, 0): 4})
    sim.run(40)                          # (before the fix rank 0 raised CvaeError here while rank 1 went on to the next collective)
    assert all(r.coop_fallback and r.fallbacks == 0 for r in sim.ranks)
    assert sim.ranks[0].skipped == sim.ranks[1].skipped == sum(1 for v in sim.reduced if v)


def test_a_second_time_out_raises_on_every_rank():
    import _cabi
    sim = Sim(lag=[0, 3], raises={(5, 0): 4, (30, 1): 4})
    with pytest.raises(_cabi.CvaeError):
        sim.run(60)
    # the rank whose host is behind takes the same decision when it gets there
    other = [r for r in sim.ranks if r._pending]
    for r in other:
        with pytest.raises(_cabi.CvaeError):
            sim.now = 100
            r.finish()


def test_two_separate_overflows_are_two_incidents():
    sim = Sim(lag=[2, 2], raises={(5, 0): 5, (40, 1): 5})
    sim.run(80)
    assert all(r.fallbacks == 2 for r in sim.ranks)


def test_a_latch_that_never_clears_is_not_mistaken_for_lagging_ranks():
    raises = {(k, 0): 5 for k in range(5, 200)}          # rank 0's kernels report an overflow in EVERY step
    sim = Sim(lag=[1, 3], raises=raises)
    sim.run(200)
    assert all(r.fallbacks >= 2 for r in sim.ranks)      # the run is cut into incidents of at most INCIDENT_MAX_STEPS steps


def test_on_one_rank_an_incident_only_covers_the_steps_that_were_in_flight():
    """without a process group nobody else can keep the word raised, so a word raised by a step enqueued AFTER the host
    handled an incident is a new incident (a persisting time-out raises at once instead of eating INCIDENT_MAX_STEPS minibatches)."""
    import _cabi
    raises = {(k, 0): 4 for k in range(5, 60)}           # a hand-off time-out in EVERY step from step 5 on
    sim = Sim(lag=[2], raises=raises)
    with pytest.raises(_cabi.CvaeError):
        sim.run(60)
    r = sim.ranks[0]
    assert r.coop_fallback                               # first incident: cooperative launches; the steps in flight continue it ...
    assert r.skipped <= 2 + 2 + 1 and sim.now < 12       # ... and the first word raised behind them raises
