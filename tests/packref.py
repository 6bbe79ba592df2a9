"""numpy statement of the packed row format of include/fmd_hip.h (fmd_ovlp_pack_dev): the checker of the pack
kernel and the producer of packed rows in the CPU (gloo) tests.  TEST INFRASTRUCTURE."""
import numpy as np

F_OVERFLOW, F_PACK4 = 2, 8


def pack_rows(rec, nei, seq, max_nei):
    """rec: OVLP_DT [n]; nei: INTV_DT [n, max_nei]; seq: uint8 [n, stride] -> (prec [n] OVLP_DT, off [n+1] uint64, var uint8)."""
    n = len(rec)
    prec = rec.copy()
    off = np.zeros(n + 1, dtype=np.uint64)
    parts = []
    pos = 0
    stride = seq.shape[1]
    for i in range(n):
        r = rec[i]
        off[i] = pos
        if r["status"] != 0 or (r["flags"] & F_OVERFLOW):
            continue
        nb = min(int(r["len"]) + int(r["ext_len"]), stride)
        s = seq[i, :nb]
        p4 = bool(((s < 1) | (s > 4)).any())
        nn = min(int(r["n_nei"]), max_nei)
        parts.append(nei[i, :nn].tobytes())
        if p4:
            prec["flags"][i] |= F_PACK4
            nbytes = (nb + 1) // 2
            b = np.zeros(2 * nbytes, dtype=np.uint8); b[:nb] = s & 15
            packed = (b[0::2] | (b[1::2] << 4)).astype(np.uint8)
        else:
            nbytes = (nb + 3) // 4
            b = np.zeros(4 * nbytes, dtype=np.uint8); b[:nb] = (s - 1) & 3
            packed = (b[0::4] | (b[1::4] << 2) | (b[2::4] << 4) | (b[3::4] << 6)).astype(np.uint8)
        pad = (-nbytes) % 8
        parts.append(packed.tobytes() + b"\0" * pad)
        pos += nn * 32 + nbytes + pad
    off[n] = pos
    var = np.frombuffer(b"".join(parts), dtype=np.uint8).copy() if parts else np.zeros(0, np.uint8)
    assert len(var) == pos
    return prec, off, var


def unpack_row(prec_row, var_row):
    """-> (neighbours INTV_DT [n_nei'], bases uint8 [len + ext_len]) of one packed row."""
    from fermi_amd.api import INTV_DT
    if len(var_row) == 0:
        return np.zeros(0, dtype=INTV_DT), np.zeros(0, np.uint8)
    nb = int(prec_row["len"]) + int(prec_row["ext_len"])
    p4 = bool(prec_row["flags"] & F_PACK4)
    sb = ((nb + 1) // 2 if p4 else (nb + 3) // 4)
    sb8 = (sb + 7) // 8 * 8
    nn = (len(var_row) - sb8) // 32
    ne = var_row[: nn * 32].view(INTV_DT)
    pk = var_row[nn * 32:]
    j = np.arange(nb)
    if p4:
        s = (pk[j // 2] >> (4 * (j % 2))) & 15
    else:
        s = ((pk[j // 4] >> (2 * (j % 4))) & 3) + 1
    return ne, s.astype(np.uint8)
