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
