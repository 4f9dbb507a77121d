"""Loads tests/golden/fixtures/*.npz (decoded reference fixtures) for both the oracle and the product."""
import os

import numpy as np

from oracle.bamio import BamData

FIXDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixtures")


def load_fixture(name: str) -> BamData:
    z = np.load(os.path.join(FIXDIR, name + ".npz"))
    names = bytes(z["ref_names"]).decode().split("\n") if z["ref_names"].size else []
    qn = bytes(z["qname"]).split(b"\n") if z["tid"].size else []
    return BamData(names, z["ref_lens"], z["tid"], z["pos"], z["flag"], z["mapq"], z["l_seq"], z["nm"],
                   z["nm_kind"], z["cigar_off"], z["cigar"], z["mtid"], z["mpos"], z["tlen"], qn, "")


def stoit(name: str) -> str:
    return os.path.splitext(name)[0]


def swap_halves(b, cut):
    """Records [cut, n) of a RecordBatch in front of records [0, cut): a file that is no longer sorted by reference."""
    import numpy as np
    from coverm_amd.engine import RecordBatch
    lo, hi = b.slice(0, cut), b.slice(cut, b.n_records)
    hi0, hi1, lo0, lo1 = int(hi.cigar_off[0]), int(hi.cigar_off[-1]), int(lo.cigar_off[0]), int(lo.cigar_off[-1])
    coff = np.concatenate([hi.cigar_off[:-1].astype(np.int64) - hi0, lo.cigar_off.astype(np.int64) - lo0 + (hi1 - hi0)]).astype(np.uint32)
    cat = lambda k: np.concatenate([getattr(hi, k), getattr(lo, k)])
    return RecordBatch(cat("tid"), cat("pos"), cat("flag"), cat("mapq"), cat("nm"), cat("nm_kind"), cat("l_seq"), coff,
                       np.concatenate([hi.cigar[hi0:hi1], lo.cigar[lo0:lo1]]))
