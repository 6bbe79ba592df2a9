"""The dealing of the get_nei kernels' work lists (fmd_deal_first / fmd_deal_init / fmd_deal_next, fermi_amd/csrc/fmd_ovlp_grp.hip) restated in Python, wave by
wave with one shared counter, and run against its contract: every list position below N goes to exactly one group, no position at or past N is ever used, every
wave stops asking (`dry`) after finitely many steps, and a list shorter than the grid is dealt one strand per group with no atomic at all -- for the group
sizes of the kernels, lists from empty to many chunks per wave, and groups that ask in random order.  (The kernels themselves are compared with the round-robin
deal and the oracle on the GPU: tests/test_gpu_parity.py::test_dealt_out_work_lists_equal_the_round_robin_deal.)"""
import random
import pytest

DEAL_CHUNK = 64          # FMD_DEAL_CHUNK


def deal_first(N, grid, G):
    g, S = N // (2 * grid), 64 // G
    return S if g < S else (DEAL_CHUNK if g > DEAL_CHUNK else g)


class Wave:
    def __init__(self, w, N, grid, G):
        self.N, self.grid, self.G, self.S = N, grid, G, 64 // G
        c0 = deal_first(N, grid, G)
        self.cur, self.end = w * c0, w * c0 + c0
        self.dry = self.cur >= N or N == 0       # (the kernels return at once in both cases)
        self.atomics = 0

    def next(self, counter, asking):
        """asking: the groups (0..S-1) whose prefetch slot is empty; -> {group: position or None}; counter: [value] of the shared word"""
        out = {g: None for g in asking}
        if self.dry or not asking:
            return out
        order = sorted(asking)                   # r = the group's place among those that ask (ballot order = lane order)
        need, avail = len(order), self.end - self.cur
        if avail < need:
            rem = self.N - self.end if self.N > self.end else 0
            gsz = rem // (2 * self.grid)
            sz = 16 if gsz < 16 else (DEAL_CHUNK if gsz > DEAL_CHUNK else gsz)
            v = counter[0] + self.grid * deal_first(self.N, self.grid, self.G)
            counter[0] += sz
            self.atomics += 1
            for r, g in enumerate(order):
                out[g] = self.cur + r if r < avail else v + (r - avail)
            self.cur, self.end = v + (need - avail), v + sz
        else:
            for r, g in enumerate(order):
                out[g] = self.cur + r
            self.cur += need
        if self.cur >= self.N:
            self.dry = True
        return out


@pytest.mark.parametrize("G", [4, 8, 12, 16, 21, 32])
@pytest.mark.parametrize("N,grid", [(0, 7), (1, 7), (5, 64), (63, 3), (64, 3), (1000, 16), (4097, 16), (20000, 33), (100000, 5), (3, 4096)])
def test_every_position_is_dealt_exactly_once(G, N, grid):
    rng = random.Random(1000 * G + N + grid)
    S = 64 // G
    waves = [Wave(w, N, grid, G) for w in range(grid)]
    counter = [0]
    got = []
    for _ in range(10 * (N + grid * S) + 100):
        live = [w for w in waves if not w.dry]
        if not live:
            break
        w = rng.choice(live)                                       # the waves run at their own pace
        asking = [g for g in range(S) if rng.random() < 0.6] or [rng.randrange(S)]
        for g, pos in w.next(counter, asking).items():
            if pos is not None and pos < N:                        # (the caller's `idx < N` test)
                got.append(pos)
    assert all(w.dry for w in waves), "a wave never stopped asking"
    assert sorted(got) == list(range(N))
    if N <= grid * S:                                              # a list shorter than the grid: one strand per group, as the round-robin deal had it
        assert all(w.atomics <= 1 for w in waves)
        assert sum(w.atomics for w in waves) <= (N + S - 1) // S   # only a wave that had work comes back (once) to learn that nothing is left


def test_first_chunks_are_the_round_robin_layout_for_short_lists():
    for G in (4, 8, 12, 16, 21, 32):
        S, grid, N = 64 // G, 100, 64 // G * 100
        waves, counter = [Wave(w, N, grid, G) for w in range(grid)], [0]
        for w, wave in enumerate(waves):
            first = wave.next(counter, list(range(S)))
            assert [first[g] for g in range(S)] == [w * S + g for g in range(S)]     # position b * S + g: the old deal's first round
        assert counter[0] == 0
