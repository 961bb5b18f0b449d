"""DESIGN.md section 7's ordering claim under a slow link: "a slow link slows the epochs down instead of corrupting
anything".  W ranks (Engine(shard=(r, W)) behind guardx_amd.dist.TapeHandoff, the default N > 1 configuration: sharded
sampler, all-tapes expansion) run in one process on one GPU; the ONE collective per epoch is played by a "link" stream
that -- like RCCL's own stream -- starts once every rank's send buffer is ready on that rank's stream, then SLEEPS for about
two epochs of device time, and only then reads the send buffers and writes every rank's receive buffer.  Work.wait() is a
stream-level wait for that, as ProcessGroupNCCL's is.  So on the device timeline the collective of epoch k is still
reading send[k % 3] and has not yet written recv[k % 3] while the host has long queued epochs k + 1, k + 2, ...; whatever
reuses those buffers (rollout_tape of epoch k + 3, the shard sampler's tail block, the collective of epoch k + 3) and the
layout pool the tape names (the sampler that recycles it) must be ordered behind it by events alone."""
import numpy as np
import pytest

from helpers import task_config

pytestmark = pytest.mark.gpu

SLEEP_CYCLES = 30_000_000          # torch.cuda._sleep: ~12-15 ms at the shader clock, many epochs' worth of these sizes


def _harness(torch, W, N, T, M, seed, sleep_cycles, lag=1, extra=None):
    from guardx_amd import Engine
    from guardx_amd.dist import TapeHandoff
    kw = dict(seed=seed, num_steps=T, goal_size=2.9, **(extra or {}))
    full = Engine(task_config(N * W, **kw), n_candidates=M)
    ranks = [Engine(task_config(N, **kw), n_candidates=M, shard=(r, W)) for r in range(W)]
    o_full = full.reset()
    for r, e in enumerate(ranks):
        assert torch.equal(e.reset(), o_full[r * N:(r + 1) * N])
        e.set_prefetch(T)
    full.set_prefetch(T)
    link = torch.cuda.Stream()
    state = {"puts": {}, "done": {}, "issued": 0}

    class Work:
        def __init__(self, ep):
            self.ep = ep

        def wait(self):                    # stream-level, as ProcessGroupNCCL: the CURRENT stream waits for the link
            torch.cuda.current_stream().wait_event(state["done"][self.ep])

    class SlowLink(TapeHandoff):
        def __init__(self, env, rank):
            super().__init__(env, T, sharded_sampler=True, expand="all", _play=(rank, W))
            self.ep = 0
            self.held = []

        def _gather(self, i, buf):
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())             # the send buffer is complete on this rank's stream
            state["puts"].setdefault(self.ep, [None] * W)[self.rank] = (buf, ev, i)
            ep, self.ep = self.ep, self.ep + 1
            return Work(ep)

        if lag > 1:                                            # the expansion enqueued `lag` epochs late (host side)
            def _expand_pending(self, install=True):
                self.held.append(self.pending)
                self.pending = None
                if len(self.held) >= lag:
                    self.pending = self.held.pop(0)
                super()._expand_pending(install)

    hs = [SlowLink(e, r) for r, e in enumerate(ranks)]

    def deliver(ep):
        """every rank has issued the collective of epoch `ep`: the link runs it -- late"""
        puts = state["puts"].pop(ep)
        with torch.cuda.stream(link):
            for _, ev, _ in puts:
                link.wait_event(ev)
            torch.cuda._sleep(sleep_cycles)
            gathered = torch.cat([b for b, _, _ in puts])      # reads every send buffer only now
            for r, h in enumerate(hs):
                h.recv[puts[r][2]].copy_(gathered)             # writes every receive buffer only now
            done = torch.cuda.Event()
            done.record(link)
        state["done"][ep] = done
    return full, ranks, hs, deliver


@pytest.mark.parametrize("W", [2, 8])
def test_slow_link_delays_the_epochs_and_corrupts_nothing(W):
    import torch
    assert torch.cuda.is_available()
    N, T, M, EPOCHS = 384, 24, 160_000, 9
    full, ranks, hs, deliver = _harness(torch, W, N, T, M, seed=23, sleep_cycles=SLEEP_CYCLES)
    rng = np.random.default_rng(4)
    refs = []
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(); torch.cuda._sleep(SLEEP_CYCLES); s1.record()
    torch.cuda.synchronize()
    sleep_ms = s0.elapsed_time(s1)
    t0.record()
    got = []                                   # (epoch, clone of every rank's expanded rows), compared after the loop:
    for ep in range(EPOCHS):                   # no host synchronisation inside, so the host runs far ahead of the link
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, N * W, 2)).astype(np.float32)).cuda()
        if ep:
            o_full = full.reset(check=False)
            for r, e in enumerate(ranks):
                o = e.reset(check=False)
                got.append(("reset", ep, r, o, o_full[r * N:(r + 1) * N]))
        *_, pk = full.rollout(acts, packed=True)
        refs.append(pk)
        for r, h in enumerate(hs):
            h.step(acts[:, r * N:(r + 1) * N].contiguous())
        deliver(ep)
        if ep >= 1:
            for r, h in enumerate(hs):
                with torch.cuda.stream(h.stream):              # behind the expansion, on the hand-off's stream
                    got.append(("rows", ep - 1, r, h.rollout.clone(), None))
    for h in hs:
        h.drain()
    t1.record()
    torch.cuda.synchronize()
    total_ms = t0.elapsed_time(t1)
    # the link throttled the pipeline: every collective slept, and they are serial on the link (_sleep counts shader
    # cycles and the calibration above ran on a colder clock than the loop: hence the margin)
    assert total_ms > 0.6 * EPOCHS * sleep_ms, (total_ms, sleep_ms)
    n_rows = 0
    for kind, ep, r, x, want in got:
        if kind == "reset":
            assert torch.equal(x, want), ("reset", ep, r)
        else:
            for s in range(W):
                w = refs[ep][:, s * N:(s + 1) * N].contiguous()
                assert torch.equal(x[s].view(torch.int32), w.view(torch.int32)), ("rows", ep, r, s)
                n_rows += 1
    assert n_rows == (EPOCHS - 1) * W * W
    for r, h in enumerate(hs):                                 # the last epoch, expanded by drain()
        for s in range(W):
            assert torch.equal(h.rollout[s].view(torch.int32),
                               refs[-1][:, s * N:(s + 1) * N].contiguous().view(torch.int32)), ("last", r, s)
    for r, e in enumerate(ranks):
        hits, misses, _ = e.prefetch_stats()
        assert hits == EPOCHS - 2 and misses == 0, (r, hits, misses)     # resets 3.. took installed pools
        assert e.check_layouts() > N * W
        np.testing.assert_array_equal(e.get_pool(64), full.get_pool(64))
    for h in hs:
        assert h.shard_skips == 0 and h.blocks_installed == EPOCHS - 2
        h.close()
    full.close()
    for e in ranks:
        e.close()


def test_an_expansion_enqueued_one_epoch_too_late_is_refused_not_wrong():
    """The guard behind the ordering: the pool a tape names is recycled by the sampler launched at the second reset
    after its rollout (three pools).  A hand-off whose expansions lag TWO epochs behind the stepping on the host (here: a
    subclass that holds each gathered epoch back one step() longer) asks for a pool whose rows are already being
    rewritten -- gx_expand_tapes refuses the stale token (GX_ERR_STATE, "resampled") instead of reading them."""
    import torch
    from guardx_amd._native import GxError, GX_ERR_STATE
    W, N, T, M = 2, 128, 12, 60_000
    full, ranks, hs, deliver = _harness(torch, W, N, T, M, seed=5, sleep_cycles=1000, lag=2)
    rng = np.random.default_rng(6)
    refused = 0
    for ep in range(5):
        acts = torch.from_numpy(rng.uniform(-1, 1, (T, N * W, 2)).astype(np.float32)).cuda()
        if ep:
            full.reset(check=False)
            for e in ranks:
                e.reset(check=False)
        full.rollout(acts)
        for r, h in enumerate(hs):
            try:
                h.step(acts[:, r * N:(r + 1) * N].contiguous())
            except GxError as exc:
                assert exc.status == GX_ERR_STATE and "resampled" in str(exc), exc
                refused += 1
                # what step() had left to do after the refused expansion: issue this epoch's collective
                i = (h.k) % h.depth
                h.works[i] = h._gather(i, h.send[i]); h.pending = None; h.k += 1
        deliver(ep)
    torch.cuda.synchronize()
    assert refused >= W, refused           # every rank saw it at least once
    for h in hs:
        h.closed = True                    # (the pipeline was broken on purpose: nothing to drain)
    full.close()
    for e in ranks:
        e.close()
