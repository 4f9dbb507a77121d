#!/usr/bin/env python3
"""Regenerates tests/golden/fixtures/*.npz from the reference's fixture BAM/SAM files.

Run in the build container, where /root/reference exists:  python tests/golden/make_golden.py
Each .npz holds the decoded records of one reference fixture (tests/data/<name>) as the SoA
the C ABI consumes (tid,pos,flag,mapq,l_seq,nm,nm_kind,cigar_off,cigar,mtid,mpos,tlen), the
header (names, lengths) and the read names, so parity tests run where the reference is absent.
Decoding uses oracle/bamio.py (pure Python BGZF/BAM/SAM reader).  Also writes the genome
definition used by the CLI goldens (same content as the reference's tests/data/7seqs.definition:
genome<TAB>contig, one line per contig, derived here from the 7seqs BAM header names).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import bamio  # noqa: E402
from tests.golden import cases  # noqa: E402

REF_DATA = "/root/reference/tests/data"


def main():
    out = os.path.join(HERE, "fixtures")
    os.makedirs(out, exist_ok=True)
    for name in cases.FIXTURE_FILES:
        d = bamio.read_alignment_file(os.path.join(REF_DATA, name))
        np.savez_compressed(
            os.path.join(out, name + ".npz"),
            ref_names=np.frombuffer("\n".join(d.ref_names).encode(), dtype=np.uint8),
            ref_lens=d.ref_lens, tid=d.tid, pos=d.pos, flag=d.flag, mapq=d.mapq, l_seq=d.l_seq, nm=d.nm,
            nm_kind=d.nm_kind, cigar_off=d.cigar_off, cigar=d.cigar, mtid=d.mtid, mpos=d.mpos, tlen=d.tlen,
            qname=np.frombuffer(b"\n".join(d.qname), dtype=np.uint8))
        print("%-45s %6d records %7d refs" % (name, d.n_records, len(d.ref_names)))
    d = bamio.read_alignment_file(os.path.join(REF_DATA, "7seqs.reads_for_seq1_and_seq2.bam"))
    with open(os.path.join(out, "7seqs.definition"), "w") as fh:
        for n in d.ref_names:
            fh.write("%s\t%s\n" % (n.split("~")[0], n))
    with open(os.path.join(out, "7seqs.definition_with_comments"), "w") as fh:   # genome_parsing.rs:189-198
        for k, n in enumerate(d.ref_names):
            fh.write("%s\t%s %s comment\n" % (n.split("~")[0], n, "a" if k == 0 else "another"))
    # per-gene goldens (genes.rs:621-766, tests/test_cmdline.rs:134-206): three gene lines as the reference's
    # tests/data/2seqs.gff describes them (gene1 = seq1:1-1000, gene2 = seq1:100-200, gene3 = seq2:1-1000, with a
    # directive, a second attribute and a comment line, which the parser must skip) and the two-line genome definition
    with open(os.path.join(out, "2seqs.gff"), "w") as fh:
        fh.write("##gff-version 3\n"
                 "seq1\ttest\tgene\t1\t1000\t.\t+\t.\tID=gene1;Name=first_gene\n"
                 "seq1\ttest\tgene\t100\t200\t.\t-\t.\tID=gene2\n"
                 "# a comment line that should be ignored\n"
                 "seq2\ttest\tgene\t1\t1000\t.\t+\t.\tID=gene3\n")
    with open(os.path.join(out, "2seqs.genome-definition"), "w") as fh:
        fh.write("genomeA\tseq1\ngenomeB\tseq2\n")
    # tests/data/genomes_dir holds seq1.fna (>seq1) and seq2.fna (>seq2): the same genome <-> contig table as a definition
    with open(os.path.join(out, "2seqs.by-file.definition"), "w") as fh:
        fh.write("seq1\tseq1\nseq2\tseq2\n")


if __name__ == "__main__":
    main()


# tests/golden/raw/*.bam: the reference's own small fixture files (htslib-written BGZF: zlib level-6 streams, its block sizes, its
# EOF markers), byte for byte — `cp /root/reference/tests/data/<name> tests/golden/raw/` for every file under 6 kB plus eg2.bam.
# tests/test_gpu_ingest.py feeds them UNMODIFIED to the device ingest on the GPU box, where /root/reference does not exist.
