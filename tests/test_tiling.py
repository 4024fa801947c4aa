"""Time tiling of the TDS encoder (tal_asrd_amd/tiling.py): the halo arithmetic against the oracle's TDS on CPU (a tile
with its halo reproduces the frames of the whole sequence), and the HIP encoder tile by tile against its one-call form."""
import numpy as np
import pytest
import torch

from oracle import tal_oracle as O
from tal_asrd_amd import synth, tiling
from tests.test_oracle_golden import _fill, _tds_shapes


def test_receptive_halo_of_the_reference_stack():
    # [-640, +780] mel frames for depths 2 / 3 / 6, total stride 8
    assert tiling.receptive_halo((2, 3, 6)) == (640, 780, 8)
    assert tiling.receptive_halo((1, 1, 2)) == (220, 360, 8)
    assert tiling.receptive_halo(()) == (0, 0, 1)


def test_out_len_matches_oracle():
    for T in (141, 160, 161, 168, 169, 3001, 30001, 360001):
        assert tiling.tds_out_len(T) == O.tds_total_out_len(T)
    assert tiling.tds_out_len(141) == 1
    for T in (0, 20, 100, 140):          # a stage gets fewer than 21 frames: nothing comes out
        assert tiling.tds_out_len(T) == 0


@pytest.mark.parametrize("T,out_tile", [(1500, 37), (1500, 1000), (2047, 64), (900, 1)])
def test_plan_covers_every_frame_once(T, out_tile):
    plan = tiling.plan_tiles(T, out_tile, (1, 1, 2))
    t_out = tiling.tds_out_len(T)
    cover = np.zeros(t_out, dtype=np.int32)
    for t in plan:
        cover[t.out_start:t.out_stop] += 1
        assert 0 <= t.in_start < t.in_stop <= T and t.in_start % 8 == 0
        assert t.skip == t.out_start - t.in_start // 8 and t.skip >= 0
        # the slice yields the frames the tile needs
        assert tiling.tds_out_len(t.in_stop - t.in_start) >= t.skip + (t.out_stop - t.out_start)
    assert (cover == 1).all()
    assert tiling.plan_tiles(100, 8, (1, 1, 2)) == []
    with pytest.raises(ValueError):
        tiling.plan_tiles(1000, 0)


@pytest.mark.parametrize("T,out_tile", [(1800, 40), (1800, 77), (1237, 25)])
def test_tiles_reproduce_the_whole_sequence_oracle(T, out_tile):
    """Every tile, run through the oracle's TDS as a sequence of its own (zero padding at ITS ends), equals the frames
    of the whole-sequence result it is responsible for -- including the tiles at the true ends."""
    sizes, depths, groups = [8, 16, 24, 32], (1, 1, 2), 8
    sd = _fill(_tds_shapes(sizes, list(depths), groups), "tds_small.")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, sizes[0], T, generator=g)
    whole = O.tds_forward(x, sd, prefix="", depths=depths, groups=groups)
    plan = tiling.plan_tiles(T, out_tile, depths)
    assert len(plan) >= 3
    for t in plan:
        y = O.tds_forward(x[:, :, t.in_start:t.in_stop], sd, prefix="", depths=depths, groups=groups)
        n = t.out_stop - t.out_start
        np.testing.assert_allclose(y[0, :, t.skip:t.skip + n].numpy(), whole[0, :, t.out_start:t.out_stop].numpy(), atol=2e-6, rtol=0)
    # and the halo is needed: with half of it the tile's first frame differs
    left, right, stride = tiling.receptive_halo(depths)
    t = plan[-1]
    cut = (t.skip // 2) * stride
    y = O.tds_forward(x[:, :, t.in_start + cut:t.in_stop], sd, prefix="", depths=depths, groups=groups)
    assert not np.allclose(y[0, :, t.skip - cut // stride].numpy(), whole[0, :, t.out_start].numpy(), atol=2e-6, rtol=0)


def test_shard_tiles_partition():
    plan = tiling.plan_tiles(30001, 256)
    parts = [tiling.shard_tiles(plan, r, 3) for r in range(3)]
    assert sorted(sum(parts, []), key=lambda t: t.out_start) == plan
    assert tiling.max_item_frames() > 4 * 360001 * 0.9      # ~3.7 h of audio per call


@pytest.mark.gpu
def test_encoder_tile_by_tile_equals_one_call(sd_weights):
    """5-minute clip: the HIP encoder over 512-frame tiles (batched 8 at a time) against its one-call form; a rank's
    share of the tiles fills exactly its frames."""
    from tests.test_gpu_parity import _load
    from tal_asrd_amd import SDModel
    model = _load(SDModel(), sd_weights)
    g = torch.Generator(device="cuda").manual_seed(3)
    mel = torch.randn(1, 30001, 80, generator=g, device="cuda")
    with torch.no_grad():
        whole = model.encoder.forward_time_major(mel)
        tiled = tiling.encode_tiled(model.encoder, mel, 512)
        assert tuple(tiled.shape) == tuple(whole.shape) == (1, 3733, 1440)
        np.testing.assert_allclose(tiled.cpu().numpy(), whole.cpu().numpy(), atol=1e-5, rtol=0)
        plan = tiling.plan_tiles(30001, 512)
        mine = tiling.shard_tiles(plan, 1, 2)
        part = tiling.encode_tiled(model.encoder, mel, 512, tiles=mine)
        for t in plan:
            ref = whole[0, t.out_start:t.out_stop] if t in mine else torch.zeros_like(whole[0, t.out_start:t.out_stop])
            np.testing.assert_allclose(part[0, t.out_start:t.out_stop].cpu().numpy(), ref.cpu().numpy(), atol=1e-5, rtol=0)


@pytest.mark.gpu
def test_long_item_goes_tile_by_tile(sd_weights):
    """An item beyond TDS.max_item_frames (the kernels' 2 GiB-per-item limit, ~3.7 h of audio; lowered here) is encoded
    tile by tile inside the ordinary entry points: encoder output and the fused head's ids equal the one-call forms."""
    from tests.test_gpu_parity import _load
    from tal_asrd_amd import SDModel, synth
    model = _load(SDModel(), sd_weights)
    wav = torch.from_numpy(synth.synth_audio(120 * 16000, seed=21))[None].cuda()
    with torch.no_grad():
        mel = model.extract_features(wav)
        whole = model.encoder.forward_time_major(mel)
        feat0, ids0, logits0 = model.speaker_ids(wav, want_logits=True)
        model.encoder.max_item_frames, model.encoder.tile_frames = 4000, 300
        try:
            tiled = model.encoder.forward_time_major(mel)
            feat1, ids1 = model.speaker_ids(wav)
        finally:
            model.encoder.max_item_frames = tiling.max_item_frames(model.encoder.sizes)
            model.encoder.tile_frames = 32768
    np.testing.assert_allclose(tiled.cpu().numpy(), whole.cpu().numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(feat1.cpu().numpy(), feat0.cpu().numpy(), atol=1e-5, rtol=0)
    top2 = torch.topk(logits0.reshape(-1, logits0.shape[-1]), 2, dim=-1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert bool((ids0.reshape(-1)[safe] == ids1.reshape(-1)[safe]).all())
