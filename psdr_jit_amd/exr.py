"""Minimal OpenEXR reader/writer for radiance images (host-side file I/O of EnvironmentMap / Bitmap, which the
reference delegates to tinyexr in src/core/bitmap_loader.cpp).  Supports what lat-long environment maps use in practice:
single-part scanline files, channels of type HALF or FLOAT (any of R, G, B, A, Y), compression NONE, ZIPS, ZIP or PIZ,
increasing-y line order.  Anything else raises.  Format: "OpenEXR File Layout" (magic 20000630, version 2)."""
import struct
import zlib

import numpy as np

_MAGIC = 20000630


def _read_cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _unpredict(raw):
    """inverse of the EXR zip predictor + interleave (ImfZip.cpp)"""
    a = np.frombuffer(raw, dtype=np.uint8).astype(np.int32)
    a[1:] -= 128
    a = np.cumsum(a, dtype=np.int64).astype(np.uint8) if a.size else a.astype(np.uint8)
    n = a.size
    out = np.empty(n, dtype=np.uint8)
    half = (n + 1) // 2
    out[0::2] = a[:half]
    out[1::2] = a[half:]
    return out.tobytes()


def read_rgb(path):
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != _MAGIC:
        raise RuntimeError("not an OpenEXR file: %s" % path)
    if version & 0x200 or version & 0x1000 or version & 0x800:
        raise RuntimeError("EXR: tiled / multi-part / deep files are not supported")
    pos = 8
    attrs = {}
    while True:
        name, pos = _read_cstr(buf, pos)
        if name == "":
            break
        typ, pos = _read_cstr(buf, pos)
        (size,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        attrs[name] = (typ, buf[pos:pos + size])
        pos += size
    channels = []
    cb = attrs["channels"][1]
    p = 0
    while cb[p] != 0:
        cname, p = _read_cstr(cb, p)
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", cb, p)
        p += 16
        if xs != 1 or ys != 1:
            raise RuntimeError("EXR: subsampled channels are not supported")
        channels.append((cname, ptype))
    comp = attrs["compression"][1][0]
    if comp not in (0, 2, 3, 4):
        raise RuntimeError("EXR: compression %d not supported (NONE, ZIPS, ZIP, PIZ only)" % comp)
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = xmax - xmin + 1, ymax - ymin + 1
    if attrs.get("lineOrder", ("", b"\0"))[1][0] not in (0, 1):
        raise RuntimeError("EXR: random line order is not supported")
    lines_per_block = {3: 16, 4: 32}.get(comp, 1)
    n_blocks = (H + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from("<%dQ" % n_blocks, buf, pos)
    bpp = {0: 4, 1: 2, 2: 4}
    dt = {0: np.uint32, 1: np.float16, 2: np.float32}
    planes = {c: np.zeros((H, W), dtype=np.float32) for c, _ in channels}
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        rows = min(lines_per_block, ymax - y + 1)
        raw_size = rows * W * sum(bpp[t] for _, t in channels)
        if comp == 4 and size < raw_size:
            from . import _psdr_core                         # the PIZ decoder is native (csrc/host/exr_piz.cpp)
            data = _psdr_core._piz_decode(bytes(data), W, rows, [bpp[t] // 2 for _, t in channels]).tobytes()
        elif comp != 0 and size < raw_size:
            data = _unpredict(zlib.decompress(data))
        q = 0
        for r in range(rows):
            for cname, t in channels:                         # channels are stored alphabetically, line by line
                n = W * bpp[t]
                planes[cname][y - ymin + r] = np.frombuffer(data, dtype=dt[t], count=W, offset=q).astype(np.float32)
                q += n
    if all(c in planes for c in "RGB"):
        return np.stack([planes["R"], planes["G"], planes["B"]], axis=-1)
    if "Y" in planes:
        return np.repeat(planes["Y"][..., None], 3, axis=-1)
    raise RuntimeError("EXR: no R,G,B or Y channels")


def write_rgb(path, img, compression="zip"):
    """float32 RGB scanline EXR (NONE or ZIP)"""
    img = np.asarray(img, dtype=np.float32)
    H, W, _ = img.shape
    comp = {"none": 0, "zip": 3}[compression]

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload

    ch = b"".join(c.encode() + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for c in "BGR") + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    head = struct.pack("<ii", _MAGIC, 2)
    head += attr("channels", "chlist", ch) + attr("compression", "compression", bytes([comp]))
    head += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
    head += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lpb = 16 if comp == 3 else 1
    n_blocks = (H + lpb - 1) // lpb
    blocks = []
    for b in range(n_blocks):
        y0 = b * lpb
        rows = min(lpb, H - y0)
        raw = b"".join(img[y0 + r, :, c].tobytes() for r in range(rows) for c in (2, 1, 0))
        if comp == 3:
            a = np.frombuffer(raw, dtype=np.uint8)
            t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
            d = t.copy()
            d[1:] = (t[1:] - t[:-1] + 128) & 255
            z = zlib.compress(d.astype(np.uint8).tobytes())
            payload = z if len(z) < len(raw) else raw
        else:
            payload = raw
        blocks.append(struct.pack("<ii", y0, len(payload)) + payload)
    table_pos = len(head)
    off = table_pos + 8 * n_blocks
    table = b""
    for blk in blocks:
        table += struct.pack("<Q", off)
        off += len(blk)
    with open(path, "wb") as fh:
        fh.write(head + table + b"".join(blocks))
