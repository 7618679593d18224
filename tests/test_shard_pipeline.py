"""GPU tests of the PIPELINED multi-GPU block (skred_shard_render_mix_pipelined: the collective and master stage of block k on
the shard's own stream beside the render of block k + 1), through the library's one-rank RCCL communicator -- the only
communicator a one-GPU box has.  Every sequence must deliver the bytes of the serial form (skred_shard_render_mix).

What these add to tests/c_shard_smoke.c (d): the block length CHANGING from call to call (the per-frame master gains of the block
in flight must not live where the next block's rows go: ADVICE r3), the kernel family changing between calls (a mid-size
enveloped bank moves between the one-voice and the two-voices-per-lane kernels with its envelope motion, and the rows of the
partial mix change their number with it), and a SERIAL call directly behind two pipelined ones (it must wait for their master
stages: commit 62292a1 of round 3, which no test told apart from its predecessor).
"""
import numpy as np
import pytest

from skred_amd import banks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from skred_amd import device
    L = device.load()
    assert L.skred_amd_device_count() > 0, "no GPU visible"
    return device


def _shard(n, bank, tables, g):
    from skred_amd import sharded
    sh = sharded.Shard(n, 0, 1)
    sh.bank.set_tables(tables)
    sh.upload(bank)
    sh.bank.set_globals(g)
    sh.init_rccl(sharded.Shard.rccl_unique_id())
    sh.set_reduce(None, always_reduce=True)            # the library's own ncclReduce, with one rank
    return sh


def _run(dev, n, recipe, plan, notes_at=()):
    """plan: list of (frames, 'p' | 's').  Returns the blocks' outputs in order (host arrays) and the kernels that rendered them."""
    import torch
    bank, tables, g = banks.RECIPES[recipe](n)
    if notes_at:                                       # every note long in its sustain stage: the bank reports "quiet" after its first block,
        e = bank["voice_amp_envelope"]                 # moves to the two-voices-per-lane kernel, and comes back with every note-on
        e["sample_start"][:] = np.uint64(g.synth_sample_count - 30000)
        e["sample_release"][:] = 0
        e["is_active"][:] = 1
    sh = _shard(n, bank, tables, g)
    st = torch.cuda.Stream()
    outs = [torch.zeros(max(f for f, _ in plan), 2, device="cuda") for _ in plan]     # one buffer per block: nothing is reused
    kernels = []
    rng = np.random.default_rng(5)
    for k, (frames, form) in enumerate(plan):
        if k in notes_at:                              # note-ons with new parameters: a control action (the bank may change kernels)
            host = bank.copy()
            ids = np.sort(rng.choice(n, 300, replace=False)).astype(np.int32)
            host["voice_phase_inc"][ids] *= np.float32(1.01)
            sh.bank.update(host, ids, dirty=1 | 256, stream=st.cuda_stream)   # SKRED_DIRTY_PARAMS | SKRED_STAMP_TRIGGER
        if form == "p":
            sh.render_mix_pipelined(frames, outs[k].data_ptr(), 2, 0, st.cuda_stream)
        else:
            sh.render_mix(frames, outs[k].data_ptr(), 2, 0, st.cuda_stream)
        kernels.append(sh.bank.last_kernel())
    sh.flush(st.cuda_stream)
    st.synchronize()
    torch.cuda.synchronize()
    res = [outs[k][:plan[k][0]].cpu().numpy().copy() for k in range(len(plan))]
    sh.close()
    return res, kernels


@pytest.mark.parametrize("n,recipe", [(3000, "c2"), (229376, "c2")])
def test_pipelined_blocks_of_changing_length_equal_the_serial_form(dev, n, recipe):
    lengths = [256, 512, 64, 512, 512, 128, 1024, 32, 512, 300, 512]
    notes = (3, 7) if n > 100000 else ()
    ser, k_ser = _run(dev, n, recipe, [(f, "s") for f in lengths], notes)
    pip, k_pip = _run(dev, n, recipe, [(f, "p") for f in lengths], notes)
    for k, (a, b) in enumerate(zip(ser, pip)):
        if k_ser[k] == k_pip[k]:
            assert (a.view(np.uint32) == b.view(np.uint32)).all(), f"block {k} ({lengths[k]} frames; kernel {k_ser[k]})"
        else:
            # which family renders a block follows the device's reports, which arrive when they arrive: the two runs may differ
            # there, and then the voices are summed in another order (same voices, same samples)
            err = np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)) / max(np.sqrt(np.mean(b.astype(np.float64) ** 2)), 1e-30)
            assert err <= 1e-5, f"block {k} ({lengths[k]} frames; kernels {k_ser[k]} / {k_pip[k]}): {err}"
    assert np.abs(ser[-1]).max() > 0
    if n > 100000:
        assert len(set(k_pip)) > 1, f"the bank was meant to change kernel families: {k_pip}"


def test_serial_call_directly_behind_pipelined_ones(dev):
    n = 20000
    plan_mixed = [(512, "p"), (512, "p"), (512, "s"), (256, "s"), (512, "p"), (512, "s")]
    ser, _ = _run(dev, n, "c2", [(f, "s") for f, _ in plan_mixed])
    mix, _ = _run(dev, n, "c2", plan_mixed)
    for k, (a, b) in enumerate(zip(ser, mix)):
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), f"block {k}"
