"""On-disk formats on either side of the hot path.

  SRS file       Params::{write, read}  poly/commitment.rs:241-294
                 u32 k | n x 32 B g (compressed) | n x 32 B g_lagrange (compressed) | u32 len | additional_data
                 The reference decompresses with `from_bytes` under a rayon `parallelize`; here the 2 x n square
                 roots run on the device (h2_dev_points_decompress) and the tables never visit the host as points.
  witness file   AssignWitnessCollection::{store_witness, fetch_witness}  helpers.rs:920-1015
                 u32 columns | column i at byte offset 4 + (i << (k + 5)): n x 32 B raw (Montgomery) Fr
                 -- consumed by create_proof_from_witness (plonk/prover.rs:916-1500)

The point encoding (x little-endian, y parity in bit 7 of byte 31, identity = zeros) is this build's convention:
pairing_bn256@30b052f is not available to check against ("parity unpinned").
"""
import struct

import numpy as np

from ._lib import check
from .prover import Params


def params_write(device, params, path, additional_data=b""):
    """Params::write.  additional_data: the compressed [s]G2 of the setup (opaque here)."""
    torch = device.torch
    with open(path, "wb") as f:
        f.write(struct.pack("<I", params.k))
        for table in (params.g, params.g_lagrange):
            with torch.cuda.stream(device.tstream):
                out = torch.empty((params.n, 32), dtype=torch.uint8, device=device.dev)
            check(device.L.h2_dev_points_compress(table.data_ptr(), params.n, out.data_ptr(), device.stream),
                  "h2_dev_points_compress")
            with torch.cuda.stream(device.tstream):
                f.write(out.cpu().numpy().tobytes())
        f.write(struct.pack("<I", len(additional_data)))
        f.write(additional_data)


def params_read(device, path):
    """Params::read -> (Params with both tables resident on the device, additional_data)"""
    torch = device.torch
    with open(path, "rb") as f:
        (k,) = struct.unpack("<I", f.read(4))
        n = 1 << k
        tables = []
        for _ in range(2):
            raw = np.frombuffer(f.read(32 * n), dtype=np.uint8)
            if raw.size != 32 * n:
                raise IOError("truncated params file")
            with torch.cuda.stream(device.tstream):
                d_raw = torch.from_numpy(raw.copy()).to(device.dev)
                pts = torch.empty((n, 8), dtype=torch.int64, device=device.dev)
            check(device.L.h2_dev_points_decompress(d_raw.data_ptr(), n, pts.data_ptr(), device.stream),
                  "h2_dev_points_decompress")
            tables.append(pts)
        (alen,) = struct.unpack("<I", f.read(4))
        additional = f.read(alen)
        if len(additional) != alen:
            raise IOError("truncated params file")
    return Params(device, k, tables[0], tables[1]), additional


def witness_store(path, k, columns):
    """store_witness: columns = (n, 4) u64 arrays in the in-memory (Montgomery) representation"""
    n = 1 << k
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(columns)))
        for col in columns:
            col = np.ascontiguousarray(col, dtype=np.uint64)
            assert col.shape == (n, 4)
            f.write(col.tobytes())


def witness_fetch(path, k):
    """fetch_witness: the columns as read-only memory maps of the file (bundle size 2^(k+5) bytes)"""
    n = 1 << k
    with open(path, "rb") as f:
        (count,) = struct.unpack("<I", f.read(4))
    return [np.memmap(path, dtype=np.uint64, mode="r", offset=4 + (i << (k + 5)), shape=(n, 4)) for i in range(count)]
