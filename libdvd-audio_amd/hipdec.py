"""ctypes binding of the C ABI in include/dvda_mlp_hip.h (batch tier).

torch is used for what it is good at here -- device memory and streams; every
compute step is the HIP library.  There is no CPU fallback: a missing library,
a missing GPU or a HIP error raises.
"""
import ctypes

import numpy as np

from . import _build

ST = dict(NO_SYNC=1 << 0, SYNC_CHANGE=1 << 1, PARITY=1 << 2, CRC=1 << 3, EOF=1 << 4, RESTART=1 << 5,
          PARAMS=1 << 6, HUFFMAN=1 << 7, FILTER=1 << 8, ENVELOPE=1 << 9, IRREGULAR=1 << 16,
          TIMING=1 << 17, MIDFRAME=1 << 18, CHAINED=1 << 19, OVERFLOW=1 << 20, TRUNCATED=1 << 21,
          CAPACITY=1 << 22, GENERAL=1 << 23, FALSE_SYNC=1 << 24, SEQ=1 << 25, COLD=1 << 26, YIELD=1 << 27)
# bits that do not invalidate the decoded PCM: the conditions the fast pass defers are informational
# once the passes behind it have decoded them (any failure there sets an error bit); a dropped
# access unit (later major sync with other stream parameters) is what the reference does too
ST_BENIGN = (ST["TRUNCATED"] | ST["CHAINED"] | ST["MIDFRAME"] | ST["TIMING"] | ST["GENERAL"] | ST["SEQ"] |
             ST["SYNC_CHANGE"] | ST["COLD"] | ST["YIELD"])


class StreamInfo(ctypes.Structure):
    _fields_ = [("mlp_frames", ctypes.c_uint64), ("pcm_frames", ctypes.c_uint64),
                ("bytes_consumed", ctypes.c_uint64), ("status", ctypes.c_uint32),
                ("channels", ctypes.c_uint32), ("substreams", ctypes.c_uint32),
                ("assignment", ctypes.c_uint32), ("group0_bps", ctypes.c_uint32),
                ("group1_bps", ctypes.c_uint32), ("group0_rate", ctypes.c_uint32),
                ("group1_rate", ctypes.c_uint32), ("segments", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


class MultiSummary(ctypes.Structure):
    """dvda_mlp_multi_summary of include/dvda_mlp_hip.h"""
    _fields_ = [("pcm_frames", ctypes.c_uint64), ("samples", ctypes.c_uint64), ("compressed_bytes", ctypes.c_uint64),
                ("compressed_bytes_max_device", ctypes.c_uint64), ("streams_with_errors", ctypes.c_uint32),
                ("devices", ctypes.c_uint32), ("device_ms_max", ctypes.c_double), ("device_ms_min", ctypes.c_double),
                ("imbalance", ctypes.c_double), ("reduction", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class HipError(RuntimeError):
    pass


_lib = None

EXPORTS = ("dvda_mlp_hip_create", "dvda_mlp_hip_destroy", "dvda_mlp_hip_index", "dvda_mlp_hip_decode",
           "dvda_mlp_hip_stream_info", "dvda_mlp_hip_segment_count", "dvda_mlp_hip_kernel_time",
           "dvda_mlp_hip_version", "dvda_mlp_hip_selftest_huff", "dvda_mlp_hip_selftest_bits",
           "dvda_mlp_hip_bounds_violations", "dvda_mlp_hip_decode_async", "dvda_mlp_hip_reserve",
           "dvda_mlp_hip_decode_time",
           "dvda_mlp_hip_set_lanes_per_segment", "dvda_mlp_hip_set_chain_form",
           "dvda_mlp_hip_set_pcm_layout", "dvda_mlp_hip_segment_info",
           "dvda_mlp_hip_segment_fir", "dvda_mlp_hip_set_initial_fir",
           "dvda_hip_open_mlpdecoder", "dvda_hip_close_mlpdecoder", "dvda_hip_mlpdecoder_decode_packet",
           "dvda_hip_mlpdecoder_status", "dvda_hip_mlpdecoder_queued_bytes", "dvda_hip_mlpdecoder_path",
           "dvda_pcm_hip_workspace_words", "dvda_pcm_hip_decode_sectors", "dvda_pcm_hip_result",
           "dvda_mlp_hip_demux_sectors", "dvda_mlp_hip_pack_wav",
           "dvda_mlp_hip_shard", "dvda_mlp_hip_create_multi", "dvda_mlp_hip_destroy_multi",
           "dvda_mlp_hip_multi_devices", "dvda_mlp_hip_decode_multi", "dvda_mlp_hip_multi_device_time")


def lib():
    """Loads (building if stale) libdvda_mlp_hip.so; raises if it cannot."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so.7; import it FIRST so this library
        # binds to the same HIP runtime (two runtimes in one process do not share
        # devices or pointers)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        import os
        so = os.environ.get("DVDA_MLP_HIP_LIB") or _build.build_hip()   # override: diagnostic A/B builds
        L = ctypes.CDLL(so)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.dvda_mlp_hip_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, u32, u32]
        L.dvda_mlp_hip_destroy.argtypes = [vp]
        L.dvda_mlp_hip_destroy.restype = None
        L.dvda_mlp_hip_index.argtypes = [vp, vp, u64, vp, vp, u32, vp]
        L.dvda_mlp_hip_decode.argtypes = [vp, vp, vp, vp, vp]
        L.dvda_mlp_hip_decode_async.argtypes = [vp, vp, vp, vp, vp]
        L.dvda_mlp_hip_reserve.argtypes = [vp, u64, u32, u32]
        L.dvda_mlp_hip_stream_info.argtypes = [vp, ctypes.POINTER(StreamInfo), u32, vp]
        L.dvda_mlp_hip_segment_count.argtypes = [vp, ctypes.POINTER(u32), vp]
        L.dvda_mlp_hip_kernel_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u32)]
        L.dvda_mlp_hip_decode_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u32)]
        L.dvda_mlp_hip_set_lanes_per_segment.argtypes = [vp, u32]
        L.dvda_mlp_hip_set_pcm_layout.argtypes = [vp, u32]
        L.dvda_mlp_hip_set_chain_form.argtypes = [vp, u32]
        L.dvda_mlp_hip_version.restype = ctypes.c_char_p
        L.dvda_hip_open_mlpdecoder.restype = vp
        L.dvda_hip_open_mlpdecoder.argtypes = [ctypes.c_uint] * 5 + [ctypes.c_int]
        L.dvda_hip_close_mlpdecoder.argtypes = [vp]
        L.dvda_hip_close_mlpdecoder.restype = None
        L.dvda_hip_mlpdecoder_decode_packet.restype = ctypes.c_uint
        L.dvda_hip_mlpdecoder_decode_packet.argtypes = [vp, vp, ctypes.c_size_t,
                                                        ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)),
                                                        ctypes.POINTER(ctypes.c_uint)]
        L.dvda_hip_mlpdecoder_status.restype = ctypes.c_uint
        L.dvda_hip_mlpdecoder_status.argtypes = [vp]
        L.dvda_hip_mlpdecoder_queued_bytes.restype = ctypes.c_size_t
        L.dvda_hip_mlpdecoder_queued_bytes.argtypes = [vp]
        L.dvda_hip_mlpdecoder_path.argtypes = [vp]
        L.dvda_pcm_hip_workspace_words.restype = ctypes.c_size_t
        L.dvda_pcm_hip_workspace_words.argtypes = [u32]
        L.dvda_pcm_hip_decode_sectors.argtypes = [vp, u32, ctypes.c_uint, ctypes.c_uint, vp, u64, vp, vp]
        L.dvda_pcm_hip_result.argtypes = [vp, u32, ctypes.POINTER(u64), ctypes.POINTER(u32), vp]
        L.dvda_mlp_hip_demux_sectors.argtypes = [vp, u32, vp, u64, vp, vp]
        L.dvda_mlp_hip_pack_wav.argtypes = [vp, u64, ctypes.c_uint, u64, ctypes.c_uint, vp, vp]
        L.dvda_mlp_hip_shard.argtypes = [vp, u32, u32, vp]
        L.dvda_mlp_hip_create_multi.argtypes = [ctypes.POINTER(vp), vp, u32, u32, u32]
        L.dvda_mlp_hip_destroy_multi.argtypes = [vp]
        L.dvda_mlp_hip_destroy_multi.restype = None
        L.dvda_mlp_hip_multi_devices.argtypes = [vp]
        L.dvda_mlp_hip_multi_devices.restype = u32
        L.dvda_mlp_hip_multi_device_time.argtypes = [vp, u32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
        L.dvda_mlp_hip_decode_multi.argtypes = [vp, vp, vp, u32, u32, vp, vp, ctypes.POINTER(StreamInfo),
                                                ctypes.POINTER(MultiSummary)]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        names = {-1: "ENODEV", -2: "ENOMEM", -3: "EINVAL", -4: "ECAPACITY", -5: "ESTATE"}
        raise HipError("%s failed: %s (%d)" % (what, names.get(rc, "?"), rc))


PCM_PLANAR, PCM_INTERLEAVED, PCM_WAV24, PCM_WAV16 = 0, 1, 2, 3      # DVDA_PCM_* of include/dvda_mlp_hip.h
CHAIN_FORM = 0      # tests: 1 / 2 force the fused / two-pass form of the chain passes on every Context made afterwards


class Context:
    """One decode context = one GPU's index workspace (dvda_mlp_hip_create)."""

    def __init__(self, device=0, max_streams=1, max_segments=1024, lanes_per_segment=0, layout=PCM_PLANAR):
        self._h = ctypes.c_void_p()
        _check(lib().dvda_mlp_hip_create(ctypes.byref(self._h), device, max_streams, max_segments),
               "dvda_mlp_hip_create")
        _check(lib().dvda_mlp_hip_set_lanes_per_segment(self._h, lanes_per_segment), "set_lanes")
        _check(lib().dvda_mlp_hip_set_pcm_layout(self._h, layout), "set_pcm_layout")
        if CHAIN_FORM:
            _check(lib().dvda_mlp_hip_set_chain_form(self._h, CHAIN_FORM), "set_chain_form")
        self.device = device
        self.n_streams = 0

    def close(self):
        if self._h:
            lib().dvda_mlp_hip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def index(self, d_bytes_ptr, total_bytes, d_off_ptr, d_len_ptr, n_streams, stream=0):
        self.n_streams = n_streams
        _check(lib().dvda_mlp_hip_index(self._h, d_bytes_ptr, total_bytes, d_off_ptr, d_len_ptr,
                                        n_streams, stream), "dvda_mlp_hip_index")

    def decode(self, d_pcm_ptr, d_out_off_ptr, d_out_stride_ptr, stream=0):
        _check(lib().dvda_mlp_hip_decode(self._h, d_pcm_ptr, d_out_off_ptr, d_out_stride_ptr, stream),
               "dvda_mlp_hip_decode")

    def decode_async(self, d_pcm_ptr, d_out_off_ptr, d_out_stride_ptr, stream=0):
        """dvda_mlp_hip_decode_async: every pass enqueued, no host wait, no allocation (see reserve)."""
        _check(lib().dvda_mlp_hip_decode_async(self._h, d_pcm_ptr, d_out_off_ptr, d_out_stride_ptr, stream),
               "dvda_mlp_hip_decode_async")

    def reserve(self, chain_pcm_frames=0, chain_segments=0, seq_streams=0):
        _check(lib().dvda_mlp_hip_reserve(self._h, chain_pcm_frames, chain_segments, seq_streams), "dvda_mlp_hip_reserve")

    def stream_info(self, n=None, stream=0):
        n = self.n_streams if n is None else n
        arr = (StreamInfo * n)()
        _check(lib().dvda_mlp_hip_stream_info(self._h, arr, n, stream), "dvda_mlp_hip_stream_info")
        return arr

    def segment_count(self, stream=0):
        v = ctypes.c_uint32()
        _check(lib().dvda_mlp_hip_segment_count(self._h, ctypes.byref(v), stream), "segment_count")
        return int(v.value)

    def decode_time(self):
        """mean device ms of a whole decode call (all passes); call before kernel_time(), which resets the ring"""
        ms = ctypes.c_double()
        n = ctypes.c_uint32()
        _check(lib().dvda_mlp_hip_decode_time(self._h, ctypes.byref(ms), ctypes.byref(n)), "decode_time")
        return float(ms.value), int(n.value)

    def kernel_time(self):
        ms = ctypes.c_double()
        n = ctypes.c_uint32()
        _check(lib().dvda_mlp_hip_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(n)), "kernel_time")
        return float(ms.value), int(n.value)


ROWS_PER_AU = {0: 40, 8: 40, 1: 80, 9: 80, 2: 160, 10: 160}


def pack_streams(streams):
    """list of uint8 arrays -> (flat uint8 array with 64 spare bytes, offsets, lengths); 16-byte aligned"""
    offs, lens, pos = [], [], 0
    for s in streams:
        offs.append(pos)
        lens.append(len(s))
        pos += (len(s) + 15) & ~15
    flat = np.zeros(pos + 64, np.uint8)
    for s, o in zip(streams, offs):
        flat[o:o + len(s)] = s
    return flat, np.asarray(offs, np.uint64), np.asarray(lens, np.uint64)


def decode_streams(streams, device=0, max_segments=None, lanes_per_segment=0, layout=PCM_PLANAR, ctx=None):
    """Decodes a list of complete MLP byte streams on the GPU.

    Returns (pcm, infos): pcm[i] is an int32 array [channels, pcm_frames] in RIFF-WAVE
    channel order -- what the reference appends to `samples` (src/mlp.c:527-533) --
    and infos[i] the dvda_mlp_stream_info of stream i.  With layout=PCM_INTERLEAVED the
    library writes frame-major (the dvda_read order) and pcm[i] is that buffer viewed
    as [pcm_frames, channels] and transposed, so callers compare the same way.  Raises
    HipError if the HIP path is unavailable; never falls back to a CPU decoder.
    `ctx`: a caller's Context to run on (it stays open and keeps its size: a batch it cannot hold raises); its lane
    and layout settings are set to this call's.
    """
    import torch
    if not torch.cuda.is_available():
        raise HipError("no GPU visible to torch: the MLP decode path is HIP-only")
    dev = torch.device("cuda", device)
    flat, offs, lens = pack_streams(streams)
    total = int(len(flat) - 64)
    if max_segments is None:
        max_segments = max(64, total // 64)
    own = ctx is None
    if own:
        ctx = Context(device, len(streams), max_segments, lanes_per_segment, layout)
    else:
        _check(lib().dvda_mlp_hip_set_lanes_per_segment(ctx._h, lanes_per_segment), "set_lanes")
        _check(lib().dvda_mlp_hip_set_pcm_layout(ctx._h, layout), "set_pcm_layout")
    try:
        d_bytes = torch.from_numpy(flat).to(dev)
        d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
        d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        ctx.index(d_bytes.data_ptr(), total, d_off.data_ptr(), d_len.data_ptr(), len(streams), st)
        try:
            ctx.segment_count(st)
        except HipError:
            if not own:
                raise
            # more major syncs (sync patterns in payload count too) than the context holds: a larger one
            v = ctypes.c_uint32()
            lib().dvda_mlp_hip_segment_count(ctx._h, ctypes.byref(v), st)
            ctx.close()
            ctx = Context(device, len(streams), int(v.value) + 64, lanes_per_segment, layout)
            ctx.index(d_bytes.data_ptr(), total, d_off.data_ptr(), d_len.data_ptr(), len(streams), st)
        infos = ctx.stream_info(stream=st)
        rows, nch = [], []
        for inf in infos:
            r = int(inf.mlp_frames) * ROWS_PER_AU.get(int(inf.group0_rate), 0)
            rows.append(r)
            nch.append(int(inf.channels))
        out_off, pos = [], 0
        for r, c in zip(rows, nch):
            out_off.append(pos)
            pos += r * c
        d_pcm = torch.zeros(max(pos, 1), dtype=torch.int32, device=dev)
        d_out_off = torch.tensor(out_off, dtype=torch.int64, device=dev)
        d_stride = torch.tensor(rows, dtype=torch.int64, device=dev)
        ctx.decode(d_pcm.data_ptr(), d_out_off.data_ptr(), d_stride.data_ptr(), st)
        infos = ctx.stream_info(stream=st)
        if any(inf.status & ST["OVERFLOW"] for inf in infos):
            # access units longer than the standard timing: the general pass reported the
            # size it needs; allocate exactly that and decode again
            rows = [max(r, int(inf.pcm_frames)) for r, inf in zip(rows, infos)]
            out_off, pos = [], 0
            for r, c in zip(rows, nch):
                out_off.append(pos)
                pos += r * c
            d_pcm = torch.zeros(max(pos, 1), dtype=torch.int32, device=dev)
            d_out_off = torch.tensor(out_off, dtype=torch.int64, device=dev)
            d_stride = torch.tensor(rows, dtype=torch.int64, device=dev)
            ctx.index(d_bytes.data_ptr(), total, d_off.data_ptr(), d_len.data_ptr(), len(streams), st)
            ctx.decode(d_pcm.data_ptr(), d_out_off.data_ptr(), d_stride.data_ptr(), st)
            infos = ctx.stream_info(stream=st)
        host = d_pcm.cpu().numpy()
        pcm = []
        for i, inf in enumerate(infos):
            r, c = rows[i], nch[i]
            if not r * c:
                a = np.zeros((c, 0), np.int32)
            elif layout == PCM_INTERLEAVED:
                a = host[out_off[i]:out_off[i] + r * c].reshape(r, c).T
            else:
                a = host[out_off[i]:out_off[i] + r * c].reshape(c, r)
            pcm.append(np.ascontiguousarray(a[:, :int(inf.pcm_frames)]))
        return pcm, list(infos)
    finally:
        if own:
            ctx.close()


def shard_c(sizes, parts):
    """dvda_mlp_hip_shard: the C restatement of shard.shard_titles -> owner[i]"""
    import numpy as np
    sz = np.ascontiguousarray(sizes, np.uint64)
    out = np.zeros(len(sz), np.uint32)
    _check(lib().dvda_mlp_hip_shard(sz.ctypes.data, len(sz), parts, out.ctypes.data), "dvda_mlp_hip_shard")
    return out


def decode_streams_multi(streams, devices, layout=PCM_PLANAR, max_segments=None):
    """dvda_mlp_hip_decode_multi: host streams dealt to the device entries of `devices` (a device may be named more
    than once), host PCM back.  -> (pcm list as decode_streams gives it, infos, MultiSummary)"""
    import numpy as np
    n = len(streams)
    bufs = [np.ascontiguousarray(s, np.uint8) for s in streams]
    lens = np.array([len(b) for b in bufs], np.uint64)
    if max_segments is None:
        max_segments = int(sum(len(b) // 2048 + 8 for b in bufs))
    devs = np.array(devices, np.int32)
    h = ctypes.c_void_p()
    _check(lib().dvda_mlp_hip_create_multi(ctypes.byref(h), devs.ctypes.data, len(devs), n, max_segments),
           "dvda_mlp_hip_create_multi")
    try:
        vb = 3 if layout == PCM_WAV24 else 2 if layout == PCM_WAV16 else 4
        caps = np.array([len(b) + 4096 for b in bufs], np.uint64)      # PCM frames: a frame takes more than a byte
        outs = [np.zeros(int(c) * 6 * vb + 16, np.uint8) for c in caps]
        sp = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bufs])
        op = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        infos = (StreamInfo * n)()
        summ = MultiSummary()
        _check(lib().dvda_mlp_hip_decode_multi(h, sp, lens.ctypes.data, n, layout, op, caps.ctypes.data, infos,
                                               ctypes.byref(summ)), "dvda_mlp_hip_decode_multi")
        pcm = []
        for i in range(n):
            f, ch, cap = int(infos[i].pcm_frames), int(infos[i].channels), int(caps[i])
            f = min(f, cap)
            if layout == PCM_PLANAR:
                pcm.append(outs[i][:cap * ch * 4].view(np.int32).reshape(ch, cap)[:, :f].copy() if ch else np.zeros((0, 0), np.int32))
            elif layout == PCM_INTERLEAVED:
                pcm.append(outs[i][:f * ch * 4].view(np.int32).reshape(f, ch).T.copy() if ch else np.zeros((0, 0), np.int32))
            else:
                pcm.append(outs[i][:f * ch * vb].copy())
        return pcm, list(infos), summ
    finally:
        lib().dvda_mlp_hip_destroy_multi(h)


def decode_streams_wav(streams, bits, device=0, lanes_per_segment=0):
    """Decodes complete MLP byte streams straight into the interleaved little-endian WAV payload dvda2wav
    writes (DVDA_PCM_WAV24 / DVDA_PCM_WAV16: the output stage fused into the decode kernels).
    -> (list of uint8 arrays, infos)."""
    import torch
    if not torch.cuda.is_available():
        raise HipError("no GPU visible to torch: the MLP decode path is HIP-only")
    assert bits in (16, 24)
    nb = bits // 8
    dev = torch.device("cuda", device)
    flat, offs, lens = pack_streams(streams)
    total = int(len(flat) - 64)
    ctx = Context(device, len(streams), max(64, total // 64), lanes_per_segment, PCM_WAV24 if bits == 24 else PCM_WAV16)
    try:
        d_bytes = torch.from_numpy(flat).to(dev)
        d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
        d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        ctx.index(d_bytes.data_ptr(), total, d_off.data_ptr(), d_len.data_ptr(), len(streams), st)
        infos = ctx.stream_info(stream=st)
        rows = [int(inf.mlp_frames) * ROWS_PER_AU.get(int(inf.group0_rate), 0) for inf in infos]
        for attempt in range(2):
            out_off, pos = [], 0
            for r, inf in zip(rows, infos):
                out_off.append(pos)
                pos += (r * int(inf.channels) * nb + 3) // 4 + 4          # int32 units, dword-aligned starts
            d_pcm = torch.zeros(max(pos, 1), dtype=torch.int32, device=dev)
            d_out_off = torch.tensor(out_off, dtype=torch.int64, device=dev)
            d_stride = torch.tensor(rows, dtype=torch.int64, device=dev)
            if attempt:
                ctx.index(d_bytes.data_ptr(), total, d_off.data_ptr(), d_len.data_ptr(), len(streams), st)
            ctx.decode(d_pcm.data_ptr(), d_out_off.data_ptr(), d_stride.data_ptr(), st)
            infos = ctx.stream_info(stream=st)
            if not any(inf.status & ST["OVERFLOW"] for inf in infos):
                break
            rows = [max(r, int(inf.pcm_frames)) for r, inf in zip(rows, infos)]
        host = d_pcm.cpu().numpy().view(np.uint8)
        out = [host[4 * o:4 * o + int(inf.pcm_frames) * int(inf.channels) * nb].copy() for o, inf in zip(out_off, infos)]
        return out, list(infos)
    finally:
        ctx.close()


class MLPDecoder:
    """Host-side mirror of the reference's mlp.h interface (src/mlp.h:29-42):

        dvda_open_mlpdecoder(parameters)         -> MLPDecoder(g0_bps, g1_bps, g0_rate, g1_rate, assignment)
        dvda_mlpdecoder_decode_packet(d, r, s)   -> d.decode_packet(bytes, samples)
        dvda_close_mlpdecoder(d)                 -> d.close()

    `samples` plays the role of the reference's aa_int: a list with one growable list/array per
    RIFF channel; decode_packet appends the PCM frames decoded by that call to every channel
    and returns their number (0 = nothing decodable yet, the bytes stay queued)."""

    def __init__(self, group_0_bps, group_1_bps, group_0_rate, group_1_rate, channel_assignment, device=0):
        self._h = lib().dvda_hip_open_mlpdecoder(group_0_bps, group_1_bps, group_0_rate, group_1_rate,
                                                 channel_assignment, device)
        if not self._h:
            raise HipError("dvda_hip_open_mlpdecoder failed (no GPU / HIP error): there is no CPU fallback")

    def decode_packet(self, data, samples):
        buf = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8)) if not isinstance(data, np.ndarray) \
            else np.ascontiguousarray(data, np.uint8)
        planar = (ctypes.POINTER(ctypes.c_int32) * 6)()
        nch = ctypes.c_uint()
        n = lib().dvda_hip_mlpdecoder_decode_packet(self._h, buf.ctypes.data if len(buf) else None, len(buf),
                                                    planar, ctypes.byref(nch))
        if n:
            for c in range(nch.value):
                samples[c].extend(np.ctypeslib.as_array(planar[c], shape=(n,)).tolist())
        return int(n)

    @property
    def status(self):
        return int(lib().dvda_hip_mlpdecoder_status(self._h))

    @property
    def queued_bytes(self):
        return int(lib().dvda_hip_mlpdecoder_queued_bytes(self._h))

    @property
    def path(self):
        """0: decoder state on the device, a call decodes its own access units; 1: batch-tier path (dvda_hip_mlpdecoder_path)"""
        return int(lib().dvda_hip_mlpdecoder_path(self._h))

    def close(self):
        if self._h:
            lib().dvda_hip_close_mlpdecoder(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pcm_decode_sectors(sectors, bits_per_sample, channels, device=0):
    """Raw-PCM AOB sectors (bytes, multiple of 2048) -> (int32 [channels, frames], bad_sectors)
    through dvda_pcm_hip_decode_sectors.  HIP only."""
    import torch
    if not torch.cuda.is_available():
        raise HipError("no GPU visible to torch: the PCM un-swizzle path is HIP-only")
    buf = np.ascontiguousarray(np.frombuffer(bytes(sectors), np.uint8)) if not isinstance(sectors, np.ndarray) \
        else np.ascontiguousarray(sectors, np.uint8)
    assert len(buf) % 2048 == 0 and len(buf)
    n = len(buf) // 2048
    dev = torch.device("cuda", device)
    d_sec = torch.from_numpy(buf).to(dev)
    cap = n * (2048 // (2 * channels * (bits_per_sample // 8))) * 2 + 2
    cap += cap & 1
    d_pcm = torch.zeros(channels * cap, dtype=torch.int32, device=dev)
    d_work = torch.zeros(int(lib().dvda_pcm_hip_workspace_words(n)), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _check(lib().dvda_pcm_hip_decode_sectors(d_sec.data_ptr(), n, bits_per_sample, channels, d_pcm.data_ptr(),
                                             cap, d_work.data_ptr(), st), "dvda_pcm_hip_decode_sectors")
    frames, bad = ctypes.c_uint64(), ctypes.c_uint32()
    _check(lib().dvda_pcm_hip_result(d_work.data_ptr(), n, ctypes.byref(frames), ctypes.byref(bad), st),
           "dvda_pcm_hip_result")
    out = d_pcm.cpu().numpy().reshape(channels, cap)[:, :frames.value].copy()
    return out, int(bad.value)


def mlp_demux_sectors(sectors, device=0):
    """AOB sectors of an MLP track -> (MLP bytes as uint8 array, bad_sectors), on the GPU."""
    import torch
    if not torch.cuda.is_available():
        raise HipError("no GPU visible to torch: the demux path is HIP-only")
    buf = np.ascontiguousarray(sectors, np.uint8)
    assert len(buf) % 2048 == 0 and len(buf)
    n = len(buf) // 2048
    dev = torch.device("cuda", device)
    d_sec = torch.from_numpy(buf).to(dev)
    d_out = torch.zeros(len(buf) + 64, dtype=torch.uint8, device=dev)
    d_work = torch.zeros(int(lib().dvda_pcm_hip_workspace_words(n)), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _check(lib().dvda_mlp_hip_demux_sectors(d_sec.data_ptr(), n, d_out.data_ptr(), len(buf), d_work.data_ptr(), st),
           "dvda_mlp_hip_demux_sectors")
    nbytes, bad = ctypes.c_uint64(), ctypes.c_uint32()
    _check(lib().dvda_pcm_hip_result(d_work.data_ptr(), n, ctypes.byref(nbytes), ctypes.byref(bad), st), "result")
    return d_out[:nbytes.value].cpu().numpy(), int(bad.value)


def pack_wav(planar, bits_per_sample, device=0):
    """int32 [channels, frames] -> interleaved little-endian WAV payload bytes, on the GPU."""
    import torch
    if not torch.cuda.is_available():
        raise HipError("no GPU visible to torch: the WAV packing path is HIP-only")
    planar = np.ascontiguousarray(planar, np.int32)
    ch, frames = planar.shape
    dev = torch.device("cuda", device)
    d_pcm = torch.from_numpy(planar).to(dev)
    d_out = torch.zeros(frames * ch * (bits_per_sample // 8) + 4, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _check(lib().dvda_mlp_hip_pack_wav(d_pcm.data_ptr(), frames, ch, frames, bits_per_sample, d_out.data_ptr(), st),
           "dvda_mlp_hip_pack_wav")
    return d_out[:frames * ch * (bits_per_sample // 8)].cpu().numpy()
