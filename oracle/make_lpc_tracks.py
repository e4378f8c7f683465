"""Realistic coefficient tracks for the sample-wise filter (TEST INFRASTRUCTURE; runs only in the build container).

VERDICT r2 #3: the conditioning machinery of csrc/lpc_ss.hip was tuned on the synthetic recipe; what do
analysis filters of real speech look like to it?  This script runs a textbook order-22 LPC analysis (autocorrelation
method, Hann window of 960 samples, hop 240 -- the frame rate and order of cfg/ae/decoder/golf-precise.yaml) over the six
ground-truth clips the reference ships (medias/samples/gt_{f1,m1}_{1,2,3}.wav, 24 kHz), cuts the tracks into 2 s utterances
of 200 frames (the benchmark shape) and stores ARRAYS ONLY: direct-form coefficients a (U, 200, 22) and gains (U, 200) =
sqrt of the prediction-error power.  No reference source text, no audio samples.

    python oracle/make_lpc_tracks.py     ->  tests/golden/g25_speech_lpc_tracks.npz
"""
import glob
import os

import numpy as np
import scipy.io.wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GOLF_REFERENCE", "/root/reference")
ORDER, WIN, HOP, FRAMES = 22, 960, 240, 200


def levinson(r, order):
    """Levinson-Durbin: autocorrelation r[0..order] -> a_1..a_order of A(z) = 1 + sum a_i z^-i, prediction error."""
    a = np.zeros(order + 1)
    a[0] = 1.0
    err = r[0]
    for i in range(1, order + 1):
        k = -(r[i] + np.dot(a[1:i], r[i - 1:0:-1])) / err
        a_prev = a.copy()
        a[1:i] = a_prev[1:i] + k * a_prev[i - 1:0:-1]
        a[i] = k
        err *= 1.0 - k * k
    return a[1:], err


def analyse(x):
    x = x.astype(np.float64)
    w = np.hanning(WIN + 1)[:-1]                       # periodic Hann
    n_frames = (len(x) - WIN) // HOP + 1
    A = np.zeros((n_frames, ORDER))
    G = np.zeros(n_frames)
    for f in range(n_frames):
        seg = x[f * HOP:f * HOP + WIN] * w
        r = np.correlate(seg, seg, "full")[WIN - 1:WIN + ORDER]
        r[0] = r[0] * (1.0 + 1e-9) + 1e-12             # white-noise floor: silent frames stay well posed
        A[f], err = levinson(r, ORDER)
        G[f] = np.sqrt(max(err, 0.0) / WIN)
    return A, G


def main():
    files = sorted(glob.glob(os.path.join(REF, "medias", "samples", "gt_*.wav")))
    assert len(files) == 6, files
    a_all, g_all, names = [], [], []
    for fn in files:
        sr, x = scipy.io.wavfile.read(fn)
        assert sr == 24000
        A, G = analyse(x)
        for u in range(A.shape[0] // FRAMES):
            a_all.append(A[u * FRAMES:(u + 1) * FRAMES])
            g_all.append(G[u * FRAMES:(u + 1) * FRAMES])
            names.append(f"{os.path.basename(fn)[:-4]}:{u}")
    out = os.path.join(ROOT, "tests", "golden", "g25_speech_lpc_tracks.npz")
    np.savez_compressed(out, a=np.stack(a_all).astype(np.float32), gain=np.stack(g_all).astype(np.float32),
                        utterance=np.array(names), hop=HOP, order=ORDER, window=WIN)
    print(out, np.stack(a_all).shape, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
