#!/usr/bin/env python3
"""tools/soak_reuse.py [N [SEED]] -- N batches of very different make-up through ONE decode context (what a long-running
caller does): batch sizes 1..48, one and two substreams, all channel assignments and rates, regular and fuzz-profile
streams, chained titles, and -- every few batches -- streams of random bytes, streams with a flipped byte, a stream cut
short.  Every stream the oracle decodes cleanly must come out identical; a stream the oracle rejects must carry a status
that is not benign or stop at the same PCM-frame count.  Looks for state that outlives a call (lane packing, chain
plans, lists, counters).

Importable: `run(n, seed)` -> (batches, streams, mismatches); tests/test_gpu_soak.py runs a fixed-seed slice of it in
the -m gpu suite (once as is, once with AMD_SERIALIZE_KERNEL=3, once on the range-checked DVDA_BOUNDS build).
Environment: SOAK_ONLY=<batch> replays one batch of the sequence on a context that has seen nothing else, SOAK_SAVE=<npz>
saves it, SOAK_WAV_EVERY=<k> decodes every k-th batch once more as a WAV payload (default 4), SOAK_DUMP=<dir> keeps
what differed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TWO = [12, 1, 0x14, 6, 9, 3, 17, 20]        # assignments the generator splits into two substreams


def batches(n, seed0, syn, hip):
    """The soak's batch sequence: yields (it, batch, meta, lanes, layout); meta[k] = (assignment, substreams,
    frames, kind) with kind 0 = random bytes, 1 = one flipped bit, 2 = cut short, else untouched."""
    rng = np.random.RandomState(seed0)
    for it in range(n):
        nb = int(rng.randint(1, 49))
        batch, meta = [], []
        for j in range(nb):
            kind = int(rng.randint(0, 20))
            S = 1 + int(rng.randint(0, 2))
            asg = int(rng.choice(TWO)) if S == 2 else int(rng.randint(0, 21))
            prof = int(rng.randint(0, 2))
            feats = 0
            if prof:
                feats = [syn.SF_FAST, syn.SF_ALL, syn.SF["CHAINED"], int(rng.randint(0, 1 << 18))][int(rng.randint(0, 4))]
            cfg = syn.make_cfg(assignment=asg, rate_code=int(rng.randint(0, 3)), n_substreams=S, n_aus=int(rng.randint(2, 70)),
                               profile=prof, features=feats, restart_interval=int(rng.randint(1, 12)))
            try:
                b, f = syn.stream(cfg, seed0 * 1000 + it * 64 + j)
            except Exception:
                continue
            if kind == 0:
                b = rng.randint(0, 256, int(rng.randint(16, 6000))).astype(np.uint8)      # random bytes
            elif kind == 1:
                b = b.copy()
                b[int(rng.randint(0, len(b)))] ^= 1 << int(rng.randint(0, 8))               # one flipped bit
            elif kind == 2:
                b = b[:int(rng.randint(8, len(b)))]                                         # cut short
            batch.append(b)
            meta.append((asg, S, f, kind))
        lanes = [0, 3, 0, 2][int(rng.randint(0, 4))]     # 0: the library's choice (small batches: the cooperative kernel), 3: the
                                                          # lane kernels picked per batch, 2: the two-wave kernel for everything
        layout = [hip.PCM_PLANAR, hip.PCM_INTERLEAVED][it & 1]
        if batch:
            yield it, batch, meta, lanes, layout


def stream_ok(hip, oracle, syn, b, asg, f, kind, got, inf):
    """Is what the HIP path returned for stream `b` what the oracle says (or flagged, where the oracle rejects)?"""
    nch = syn.channels(asg)
    if int(inf.channels) and int(inf.channels) != nch:
        nch = int(inf.channels)         # (a flipped bit in the first major sync's channel assignment: the stream says so)
    want, r, st = oracle.decode(b, nch, max(f, 1) + 4000)
    if st == 0 and r == 0:
        ok = got.shape[1] == 0          # nothing decodable (no major sync, or not one whole access unit): DVDA_ST_NO_SYNC is fine
    elif st == 0 and kind in (1, 2) and (inf.status & ~hip.ST_BENIGN) != 0:
        # a damaged stream the reference happens to get through without noticing (a flipped bit in a frame's
        # size field makes it swallow the frames behind it, then it stalls on what follows): the index does not
        # trust a size chain that lands nowhere and says so -- reported is as good as decoded
        ok = True
    elif st == 0:
        ok = (inf.status & ~hip.ST_BENIGN) == 0 and got.shape[1] == r and np.array_equal(got, want[:, :r])
    else:
        ok = (inf.status & ~hip.ST_BENIGN) != 0 or (got.shape[1] == r and np.array_equal(got, want[:, :r]))
    return ok, want, r, st


def run(n=300, seed0=4242, verbose=True):
    import libdvd_audio_amd as pkg
    from tests import oracle_lib
    oracle = oracle_lib.Oracle()
    syn, hip = pkg.synth, pkg.hipdec
    ctx = hip.Context(0, 64, 1 << 15)
    bad = cases = done = 0
    wav_every = int(os.environ.get("SOAK_WAV_EVERY", "4"))
    for it, batch, meta, lanes, layout in batches(n, seed0, syn, hip):
        if os.environ.get("SOAK_ONLY") and it != int(os.environ["SOAK_ONLY"]):
            continue                    # (replay of one batch of the sequence, on a context that has seen nothing else)
        if os.environ.get("SOAK_ONLY") and os.environ.get("SOAK_SAVE"):
            np.savez_compressed(os.environ["SOAK_SAVE"], n=len(batch), lanes=lanes, layout=layout,
                                **{"s%d" % i: x for i, x in enumerate(batch)})
        done += 1
        pcm, infos = hip.decode_streams(batch, lanes_per_segment=lanes, layout=layout, ctx=ctx)
        wav = None
        if it % wav_every == 0:
            # the same batch once more with the WAV payload as the output layout (a context of its own)
            bits = [24, 16][(it // 4) & 1]
            wav = (bits,) + hip.decode_streams_wav(batch, bits, lanes_per_segment=lanes)
        for k, (b, (asg, S, f, kind), got, inf) in enumerate(zip(batch, meta, pcm, infos)):
            if wav is not None and (inf.status & ~hip.ST_BENIGN) == 0 and got.shape[1]:
                payload = wav[1][k]
                winf = wav[2][k]
                if (winf.status & ~hip.ST_BENIGN) != 0 or payload.tobytes() != oracle.wav_pack(got, wav[0]):
                    bad += 1
                    print("WAV MISMATCH batch %d stream %d asg %d S %d bits %d st %#x" % (it, k, asg, S, wav[0], winf.status), flush=True)
                    # what differs, and does the same call say the same thing again (a race or stale memory does not)
                    exp = np.frombuffer(oracle.wav_pack(got, wav[0]), np.uint8)
                    pay = np.frombuffer(payload.tobytes(), np.uint8)
                    m = min(exp.size, pay.size)
                    d = np.nonzero(exp[:m] != pay[:m])[0]
                    spf = syn.channels(asg) * wav[0] // 8
                    print("   bytes %d vs %d; %d differ, frames %s .. %s of %d" % (pay.size, exp.size, d.size,
                          d[0] // spf if d.size else -1, d[-1] // spf if d.size else -1, got.shape[1]), flush=True)
                    for t in range(3):
                        again = hip.decode_streams_wav(batch, wav[0], lanes_per_segment=lanes)
                        print("   again %d: %s" % (t, "same as expected" if again[0][k].tobytes() == exp.tobytes() else
                                                     "same wrong bytes" if again[0][k].tobytes() == pay.tobytes() else "other bytes"), flush=True)
                    if os.environ.get("SOAK_DUMP"):
                        os.makedirs(os.environ["SOAK_DUMP"], exist_ok=True)
                        np.savez_compressed(os.path.join(os.environ["SOAK_DUMP"], "wav_%d_%d_%d.npz" % (seed0, it, k)), mlp=b,
                                            got=got, pay=pay, exp=exp, lanes=lanes, bits=wav[0])
            cases += 1
            ok, want, r, st = stream_ok(hip, oracle, syn, b, asg, f, kind, got, inf)
            if not ok:
                bad += 1
                if os.environ.get("SOAK_DUMP"):
                    os.makedirs(os.environ["SOAK_DUMP"], exist_ok=True)
                    np.savez_compressed(os.path.join(os.environ["SOAK_DUMP"], "reuse_%d_%d_%d.npz" % (seed0, it, bad)), mlp=b,
                                        got=got, want=want[:, :r], meta=np.array([asg, S, f, st, r, inf.status], np.int64))
                print("MISMATCH batch %d asg %d S %d oracle st %#x r %d | hip st %#x frames %d" % (it, asg, S, st, r, inf.status,
                                                                                                  got.shape[1]), flush=True)
    ctx.close()
    if verbose:
        print("soak_reuse: %d batches, %d streams, %d mismatches" % (done, cases, bad), flush=True)
    return done, cases, bad


if __name__ == "__main__":
    n_ = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed_ = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
    sys.exit(1 if run(n_, seed_)[2] else 0)
