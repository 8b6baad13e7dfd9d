"""dvda_mlp_hip_decode_async + dvda_mlp_hip_reserve: the non-blocking decode call (no host wait, no allocation) gives
what the blocking call gives when the reservation covers the batch, and says DVDA_ST_CAPACITY -- never short or wrong
PCM under a clean status -- when it does not."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _decode(pkg, streams, reserve, lanes=0, layout=None):
    """index + decode_async on a fresh context with `reserve` = (pcm_frames, segments, seq_streams) or None"""
    import torch
    hip = pkg.hipdec
    layout = hip.PCM_PLANAR if layout is None else layout
    dev = torch.device("cuda", 0)
    flat, offs, lens = hip.pack_streams(streams)
    total = int(len(flat) - 64)
    ctx = hip.Context(0, len(streams), max(64, total // 64), lanes, layout)
    try:
        if reserve is not None:
            ctx.reserve(*reserve)
        d_bytes = torch.from_numpy(flat).to(dev)
        d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
        d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        ctx.index(d_bytes.data_ptr(), total, d_off.data_ptr(), d_len.data_ptr(), len(streams), st)
        infos = ctx.stream_info(stream=st)
        rows = [int(i.mlp_frames) * hip.ROWS_PER_AU.get(int(i.group0_rate), 0) for i in infos]
        nch = [int(i.channels) for i in infos]
        out_off = np.concatenate([[0], np.cumsum([r * c for r, c in zip(rows, nch)])[:-1]]).astype(np.int64)
        d_pcm = torch.zeros(max(int(sum(r * c for r, c in zip(rows, nch))), 1), dtype=torch.int32, device=dev)
        d_out_off = torch.from_numpy(out_off).to(dev)
        d_stride = torch.tensor(rows, dtype=torch.int64, device=dev)
        ctx.decode_async(d_pcm.data_ptr(), d_out_off.data_ptr(), d_stride.data_ptr(), st)
        infos = ctx.stream_info(stream=st)
        host = d_pcm.cpu().numpy()
        pcm = []
        for i, inf in enumerate(infos):
            r, c = rows[i], nch[i]
            a = host[out_off[i]:out_off[i] + r * c]
            a = a.reshape(r, c).T if layout == hip.PCM_INTERLEAVED else a.reshape(c, r)
            pcm.append(np.ascontiguousarray(a[:, :int(inf.pcm_frames)]))
        return pcm, list(infos)
    finally:
        ctx.close()


@pytest.mark.parametrize("lanes", [0, 1, 64])
def test_regular_titles_need_no_reservation(pkg, oracle, lanes):
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=40)
    streams = [syn.stream(cfg, 6100 + i) for i in range(6)]
    pcm, infos = _decode(pkg, [b for b, _ in streams], None, lanes=lanes, layout=hip.PCM_INTERLEAVED)
    for (b, f), got, inf in zip(streams, pcm, infos):
        want, r, st = oracle.decode(b, 6, f)
        assert st == 0 and inf.status == 0 and inf.pcm_frames == f and np.array_equal(got, want)


@pytest.mark.parametrize("S", [1, 2])
def test_chained_and_deferred_titles_on_reserved_workspaces(pkg, oracle, S):
    syn, hip = pkg.synth, pkg.hipdec
    SF = syn.SF
    cfgs = [syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=32, profile=1,
                         features=SF["CHAINED"] | SF["FIRRAND"], restart_interval=4),
            syn.make_cfg(assignment=12 if S == 2 else 1, rate_code=0, n_substreams=S, n_aus=24, profile=1,
                         features=SF["MIDMATRIX"] | SF["PARAMBLOCKS"] | SF["MATRIXRAND"] | SF["VARBLOCK"], restart_interval=3),
            syn.make_cfg(assignment=12, rate_code=2, n_substreams=S, n_aus=20, profile=1,
                         features=SF["VARROWS"] | SF["VARBLOCK"], restart_interval=5),
            syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=24)]
    streams = [syn.stream(c, 6200 + 10 * S + i) for i, c in enumerate(cfgs)]
    total_rows = sum(f for _, f in streams) + 4000
    for lanes in (S, 0):
        pcm, infos = _decode(pkg, [b for b, _ in streams], (total_rows, 64, 8), lanes=lanes)
        for (b, f), got, inf in zip(streams, pcm, infos):
            want, r, st = oracle.decode(b, 6 if got.shape[0] == 6 else got.shape[0], f)
            assert st == 0 and r == f
            if inf.status & hip.ST["OVERFLOW"]:
                continue            # (non-standard timing: the rows do not fit the standard-length buffer; reported)
            assert (inf.status & ~hip.ST_BENIGN) == 0 and inf.pcm_frames == f, hex(inf.status)
            assert np.array_equal(got, want)


def test_too_small_a_reservation_is_reported_not_decoded_short(pkg, oracle):
    syn, hip = pkg.synth, pkg.hipdec
    SF = syn.SF
    chained = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=32, profile=1,
                           features=SF["CHAINED"] | SF["FIRRAND"], restart_interval=4)
    plain = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=32)
    streams = [syn.stream(chained, 6301), syn.stream(plain, 6302), syn.stream(chained, 6303)]
    for reserve in (None, (80, 1, 0)):
        pcm, infos = _decode(pkg, [b for b, _ in streams], reserve, lanes=1)
        for k, ((b, f), got, inf) in enumerate(zip(streams, pcm, infos)):
            want, r, st = oracle.decode(b, 6, f)
            if k == 1:
                assert inf.status == 0 and np.array_equal(got, want)
            else:
                assert inf.status & hip.ST["CAPACITY"], hex(inf.status)
    # and the blocking call (which sizes its workspaces itself) decodes the same batch
    pcm, infos = hip.decode_streams([b for b, _ in streams], lanes_per_segment=1)
    for (b, f), got, inf in zip(streams, pcm, infos):
        want, r, st = oracle.decode(b, 6, f)
        assert (inf.status & ~hip.ST_BENIGN) == 0 and np.array_equal(got, want)
