"""DeviceReplay ring semantics on the CPU device (host logic of the batched DeepQ agent)."""
import torch

from safe_grid_agents_amd.deepq_batched import DeviceReplay


def test_ring_overwrites_oldest_slice_and_samples_only_filled():
    n, nc = 5, 4
    rp = DeviceReplay(n, nc, slices=3, device="cpu")
    assert len(rp) == 0
    for k in range(4):
        s = torch.full((n, nc), k, dtype=torch.int8)
        rp.add_slice(s, torch.full((n,), k, dtype=torch.uint8), torch.full((n,), -k, dtype=torch.int8), s + 1,
                     torch.zeros(n, dtype=torch.bool))
        assert len(rp) == min(k + 1, 3) * n
    assert rp.head == 1 and sorted(rp.states[:, 0, 0].tolist()) == [1, 2, 3]  # slice 0 was overwritten by k = 3
    torch.manual_seed(0)
    st, a, r, su, term = rp.sample(256)
    assert st.shape == (256, nc) and set(a.tolist()) == {1, 2, 3}
    assert ((su - st) == 1).all() and (r.to(torch.int64) == -a.to(torch.int64)).all() and not term.any()


def test_partial_fill_never_samples_unwritten_slices():
    rp = DeviceReplay(3, 2, slices=8, device="cpu")
    s = torch.ones((3, 2), dtype=torch.int8)
    rp.add_slice(s, torch.ones(3, dtype=torch.uint8) * 2, torch.ones(3, dtype=torch.int8), s, torch.ones(3, dtype=torch.bool))
    st, a, r, su, term = rp.sample(100)
    assert (a == 2).all() and term.all()
