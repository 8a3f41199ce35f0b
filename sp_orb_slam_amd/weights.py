"""Weight interchange for the SuperPoint front-end.

The reference loads a libtorch archive keyed by the register_module names
(/root/reference/orb_slam2/src/cv/sp_extractor.cpp:46-62,355):
conv{1a,1b,2a,2b,3a,3b,4a,4b,Pa,Pb,Da,Db}.{weight,bias}, OIHW fp32,
1,300,865 parameters.  libspfe takes the same tensors as ONE flat fp32 blob in
that order (weight then bias per layer); on disk the blob is preceded by a
16-byte header {b"SPFW", u32 version=1, u64 num_params}.

`superpoint.pt` is absent from the reference snapshot
(.MISSING_LARGE_BLOBS:12), so tests and benchmarks use seeded synthetic
weights made here (numpy PCG64: bit-identical on every machine).
"""
import struct

import numpy as np

# (name, cin, cout, ksize) — sp_extractor.cpp:16-43
LAYERS = [
    ("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3),
    ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3),
    ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 128, 256, 3), ("convPb", 256, 65, 1),
    ("convDa", 128, 256, 3), ("convDb", 256, 256, 1),
]
NUM_PARAMS = sum(co * ci * k * k + co for _, ci, co, k in LAYERS)  # 1,300,865
MAGIC = b"SPFW"


def layer_slices():
    """name -> (weight slice, weight shape, bias slice) into the flat blob."""
    out, off = {}, 0
    for name, ci, co, k in LAYERS:
        nw = co * ci * k * k
        out[name] = (slice(off, off + nw), (co, ci, k, k), slice(off + nw, off + nw + co))
        off += nw + co
    return out


def from_named_tensors(named):
    """Dict {'conv1a.weight': array OIHW, 'conv1a.bias': array, ...} -> flat blob."""
    blob = np.empty(NUM_PARAMS, np.float32)
    for name, (ws, shape, bs) in layer_slices().items():
        w = np.asarray(named[name + ".weight"], np.float32)
        b = np.asarray(named[name + ".bias"], np.float32)
        if tuple(w.shape) != shape or b.shape != (shape[0],):
            raise ValueError("bad shape for %s: %s / %s" % (name, w.shape, b.shape))
        blob[ws] = w.reshape(-1)
        blob[bs] = b
    return blob


def to_named_tensors(blob):
    blob = np.asarray(blob, np.float32)
    if blob.size != NUM_PARAMS:
        raise ValueError("blob has %d params, expected %d" % (blob.size, NUM_PARAMS))
    out = {}
    for name, (ws, shape, bs) in layer_slices().items():
        out[name + ".weight"] = blob[ws].reshape(shape).copy()
        out[name + ".bias"] = blob[bs].copy()
    return out


def save(path, blob):
    blob = np.ascontiguousarray(blob, np.float32)
    if blob.size != NUM_PARAMS:
        raise ValueError("blob has %d params, expected %d" % (blob.size, NUM_PARAMS))
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<IQ", 1, blob.size))
        f.write(blob.tobytes())


def load(path):
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) != 16 or head[:4] != MAGIC:
            raise ValueError("%s: not an SPFW weight file" % path)
        ver, n = struct.unpack("<IQ", head[4:])
        if ver != 1 or n != NUM_PARAMS:
            raise ValueError("%s: version %d / %d params unsupported" % (path, ver, n))
        blob = np.frombuffer(f.read(4 * n), np.float32).copy()
    if blob.size != n:
        raise ValueError("%s: truncated" % path)
    return blob


def synthetic(seed=7, detector="dense"):
    """Seeded He-normal weights (fan-in), small biases.

    detector="dense": every cell passes the 0.007 threshold (1/65 > 0.007) —
        the worst case for the selection stage.
    detector="sparse": the dustbin logit gets a positive bias and the 64
        position logits a larger gain, so only ~1/4 of the cells yield a
        candidate and many pixels fall under the 0.001 heat floor — closer to a
        trained detector's statistics.
    detector="trackable": the dustbin logit is a CONSTANT (zero weights, bias 14)
        and the position logits get a gain of 10, so dense_dust = 1 - sum of the
        position probabilities is low exactly where some position logit is high —
        what a trained SuperPoint's dustbin means, and what the tracker's direct
        alignment (optimizer_dust.cpp:170-294) needs from the map: dust ~0.47 at
        keypoint cells against ~0.79 on average.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    named = {}
    for name, ci, co, k in LAYERS:
        std = np.sqrt(2.0 / (ci * k * k))
        named[name + ".weight"] = (rng.standard_normal((co, ci, k, k)) * std).astype(np.float32)
        named[name + ".bias"] = (rng.standard_normal(co) * 0.05).astype(np.float32)
    if detector == "sparse":
        named["convPb.weight"] = named["convPb.weight"] * np.float32(3.5)
        named["convPb.bias"][64] += np.float32(7.75)
    elif detector == "trackable":
        named["convPb.weight"] = named["convPb.weight"] * np.float32(10.0)
        named["convPb.weight"][64] = 0
        named["convPb.bias"][64] = np.float32(14.0)
    elif detector != "dense":
        raise ValueError("detector must be 'dense', 'sparse' or 'trackable'")
    return from_named_tensors(named)
