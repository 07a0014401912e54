"""Kaldi-style ``feats.scp`` values the reference's loaders accept (articulatory/datasets/scp_dataset.py:20-46, utils/utils.py:177-291):

    utt_id /path/utt_id.npy              numpy file
    utt_id /path/utt_id.h5[:dataset]     HDF5 file, dataset "feats" unless named (HDF5ScpLoader)
    utt_id /path/feats.ark:12345         Kaldi archive, byte offset of the matrix (kaldiio.load_scp in the reference)

kaldiio is not a dependency here: binary Kaldi float / double matrices and vectors ("FM ", "DM ", "FV ", "DV ") are parsed directly;
compressed matrices ("CM", "CM2", "CM3") raise."""
import struct

import numpy as np


def read_kaldi_matrix(path, offset):
    with open(path, "rb") as f:
        f.seek(offset)
        if f.read(2) != b"\0B":
            raise ValueError(f"{path}:{offset}: not a binary Kaldi object (text archives are not supported)")
        token = b""
        while not token.endswith(b" "):
            c = f.read(1)
            if not c or len(token) > 8:
                raise ValueError(f"{path}:{offset}: bad Kaldi header")
            token += c
        token = token.strip().decode()
        if token not in ("FM", "DM", "FV", "DV"):
            raise ValueError(f"{path}:{offset}: Kaldi object type {token!r} is not supported (compressed matrices need kaldiio)")
        dtype = np.dtype("<f4" if token[0] == "F" else "<f8")

        def dim():
            (size,) = struct.unpack("<b", f.read(1))
            if size != 4:
                raise ValueError(f"{path}:{offset}: bad dimension field")
            return struct.unpack("<i", f.read(4))[0]

        if token[1] == "M":
            rows, cols = dim(), dim()
            return np.frombuffer(f.read(rows * cols * dtype.itemsize), dtype).reshape(rows, cols).copy()
        n = dim()
        return np.frombuffer(f.read(n * dtype.itemsize), dtype).copy()


def load_scp_value(value, default_dataset="feats"):
    """The array an scp value names."""
    if ":" in value:
        path, tail = value.rsplit(":", 1)
        if path.endswith(".ark"):
            return read_kaldi_matrix(path, int(tail))
        if path.endswith(".h5"):
            from .hdf5 import read_hdf5

            return read_hdf5(path, tail)
        raise ValueError("Not supported feats.scp type.")
    if value.endswith(".h5"):
        from .hdf5 import read_hdf5

        return read_hdf5(value, default_dataset)
    if value.endswith(".npy"):
        return np.load(value)
    raise ValueError("Not supported feats.scp type.")


def is_supported(value):
    head = value.rsplit(":", 1)[0] if ":" in value else value
    return head.endswith((".npy", ".h5", ".ark")) and (":" in value or not head.endswith(".ark"))
