"""Checkpoint-format compatibility on the CPU: files in the reference's dictionary layout, written from
reference modules by oracle/make_golden.py (current layout and a synthesised old layout), must load into the product's nets
exactly as the reference's own from_state_dict loads them (expected tensors recorded next to the files)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, split_state


@pytest.mark.parametrize("tag", ["new", "old"])
def test_from_state_dict_matches_reference_loader(ngan, tag):
    exp = load_golden("ref_checkpoint_expected")
    path = os.path.join(GOLDEN, f"ref_checkpoint_{tag}.pth")
    G = ngan.models.Generator_PG.from_state_dict(path, verbose=False)
    D = ngan.models.Discriminator_PG.from_state_dict(path, verbose=False)
    meta = exp[f"{tag}/meta"]
    assert (G.image_size, D.image_size) == (int(meta[0]), int(meta[2]))
    assert abs(G.alpha_value() - meta[1]) < 1e-7 and abs(D.alpha_value() - meta[3]) < 1e-7
    for net, prefix in ((G, f"{tag}/G/"), (D, f"{tag}/D/")):
        want = split_state(exp, prefix)
        got = net.state_dict()
        assert list(got.keys()) == list(want.keys())
        for k, v in want.items():
            assert np.array_equal(got[k].numpy(), v), k


def test_checkpointer_writes_the_reference_layout(ngan, tmp_path):
    ref = ngan.utils.load_checkpoint_dict(os.path.join(GOLDEN, "ref_checkpoint_new.pth"))
    G = ngan.models.Generator_PG([16, 16, 16], image_size_init=4)
    D = ngan.models.Discriminator_PG([16, 16, 16], image_size_init=4)
    G.set_resolution(8, 1.0)
    D.set_resolution(8, 1.0)
    f = str(tmp_path / "GenDisc_t.pth")
    ck = ngan.utils.Checkpointer(G, D, 1e-4, f, N_epochs=20, verbose=False, extra_checkpoint_period=1e3)
    ck.Loss_real[:5] = np.arange(5.0)
    ck.save_state(5)
    mine = ngan.utils.load_checkpoint_dict(f)
    assert set(ref.keys()) <= set(mine.keys())                      # same keys (an optimizer_state key may be added)
    assert set(mine["Generator_attrs"]) == set(ref["Generator_attrs"]) and set(mine["Discriminator_attrs"]) == set(ref["Discriminator_attrs"])
    assert list(mine["Generator_state"].keys()) == list(ref["Generator_state"].keys())
    assert list(mine["Discriminator_state"].keys()) == list(ref["Discriminator_state"].keys())
    assert mine["epoch"] == 5 and len(mine["Loss_real"]) == 5
    # resume into fresh nets
    G2 = ngan.models.Generator_PG([16, 16, 16], image_size_init=4)
    D2 = ngan.models.Discriminator_PG([16, 16, 16], image_size_init=4)
    ck2 = ngan.utils.Checkpointer(G2, D2, 1e-4, f, N_epochs=20, verbose=False)
    ck2.load_state()
    assert ck2.epoch == 5 and G2.image_size == 8 and list(ck2.Loss_real[:5]) == [0, 1, 2, 3, 4]
    for a, b in zip(G.state_dict().values(), G2.state_dict().values()):
        assert torch.equal(a, b)
    # weights-only load from another file (the reference's --weights_init path)
    G3 = ngan.models.Generator_PG([16, 16, 16], image_size_init=4)
    D3 = ngan.models.Discriminator_PG([16, 16, 16], image_size_init=4)
    ck3 = ngan.utils.Checkpointer(G3, D3, 1e-4, str(tmp_path / "other.pth"), N_epochs=20, verbose=False)
    ck3.load_state(os.path.join(GOLDEN, "ref_checkpoint_new.pth"))
    assert ck3.epoch == 0 and G3.image_size == 8
    assert torch.equal(G3.state_dict()["layers.0.weight"], ref["Generator_state"]["layers.0.weight"])


def test_checkpoint_loader_executes_nothing_from_the_file(ngan, tmp_path):
    """Checkpoints are loaded with torch's weights-only unpickler plus an explicit numpy allow-list (the reference's files hold
    numpy loss series).  A pickle that wants to call anything else is refused, not run."""
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("touch " + str(tmp_path / "pwned"),))
    f = str(tmp_path / "evil.pth")
    torch.save({"epoch": 1, "payload": Evil()}, f)
    with pytest.raises(pickle.UnpicklingError):
        ngan.utils.load_checkpoint_dict(f)
    assert not os.path.exists(str(tmp_path / "pwned"))
    G = ngan.models.Generator_PG([16, 16, 16], image_size_init=4)
    with pytest.raises(pickle.UnpicklingError):
        type(G).from_state_dict(f, verbose=False)
    assert not os.path.exists(str(tmp_path / "pwned"))


def test_image_grid_and_cli_parser(ngan):
    grid = ngan.utils.make_image_grid(torch.rand(5, 1, 4, 4), nrow=2)
    assert tuple(grid.shape) == (1, 3 * 6 + 2, 2 * 6 + 2)
    p = ngan.train.build_arg_parser()
    o = p.parse_args(["--pggan", "--grad_pen_lambda", "10", "--batch_size", "4", "--ID", "0042"])
    assert o.pggan and o.grad_pen_lambda == 10 and o.batch_size == 4 and o.ID == "0042"
    ds = ngan.train.TensorImageDataset.synthetic(6, 16)
    ds.set_image_size(4)
    assert tuple(ds[0].shape) == (1, 4, 4) and len(ds) == 6
    assert torch.allclose(ds[0], torch.nn.functional.avg_pool2d(ds.full[0:1], 4)[0], atol=1e-6)


def test_seeded_sampling_latents_follow_the_reference(ngan):
    """gen_samples' latent draw (reference utils.py:346-355 over 57-92) against tests/golden/sampling_small.npz, which
    oracle/make_golden.py captured over the reference's Generator_PG: same vectors bit for bit, global RNG stream untouched,
    second call served from the memo."""
    from conftest import load_golden
    fix = load_golden("sampling_small")
    res, seed, n, size_max, init, latent = (int(v) for v in fix["meta"])
    ngan.utils.Latent_vecs_memo.clear()
    torch.manual_seed(1234)
    before = torch.get_rng_state()
    z = ngan.utils.sample_latent_vec((n, latent), seed=seed)
    assert torch.equal(before, torch.get_rng_state())
    assert np.array_equal(z.numpy(), fix["z"])
    assert ngan.utils.sample_latent_vec((n, latent), seed=seed) is not None and ((n, latent), "randn", seed) in ngan.utils.Latent_vecs_memo
    # the grid: nearest-neighbour enlargement (utils.py:598-601) laid out row-major with the normalisation of save_image(normalize=True)
    images = torch.from_numpy(fix["images"])
    enlarged = torch.nn.functional.interpolate(images, size=(size_max, size_max))
    assert np.array_equal(enlarged.numpy(), fix["enlarged"])
    grid = ngan.utils.make_image_grid(enlarged, nrow=int(np.round(np.sqrt(n))))
    lo, hi = float(enlarged.min()), float(enlarged.max())
    tile = grid[0, 2:2 + size_max, 2 + (size_max + 2):2 + (size_max + 2) + size_max]          # second image of the first row
    assert torch.allclose(tile, (enlarged[1, 0] - lo) / (hi - lo), atol=1e-6)
