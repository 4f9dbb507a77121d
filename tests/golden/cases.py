"""Golden vectors for the BAM -> pileup -> per-contig / per-genome path.

Every expected string below is the literal expectation asserted by one of the reference's own
tests (wwood/CoverM v0.8.0); `cite` gives file:line.  Inputs are the reference's fixture BAMs,
decoded once into tests/golden/fixtures/*.npz by tests/golden/make_golden.py (the GPU box has
no /root/reference).

Estimator tuples mirror the reference constructors (estimators.rs:107-224):
  ("mean", min_frac, excl, exclude_mismatches)      ("variance", min_frac, excl)
  ("trimmed_mean", min, max, min_frac, excl)        ("pileup_counts", min_frac, excl)
  ("covered_fraction", min_frac) ("covered_bases", min_frac) ("rpkm", min_frac) ("tpm", min_frac)
  ("length",) ("read_count",) ("reads_per_base",) ("anir",)

flag filter tuple = (include_improper_pairs, include_supplementary, include_secondary).
"""

S7 = "7seqs.reads_for_seq1_and_seq2"
GECO_SE = (["se"], {"seq1": 0, "seq2": 0})
GECO_S = (["s"], {"seq1": 0, "seq2": 0})
GECO_7 = (["genome1", "genome2", "genome3", "genome4", "genome5", "genome6"], {
    "genome1~random_sequence_length_11000": 0, "genome1~random_sequence_length_11010": 0,
    "genome2~seq1": 1, "genome3~random_sequence_length_11001": 2,
    "genome4~random_sequence_length_11002": 3, "genome5~seq2": 4,
    "genome6~random_sequence_length_11003": 5})
GECO_23 = (["genome2", "genome3"], {"genome2~seq1": 0, "genome3~random_sequence_length_11001": 1})

# harness flag filters
FF_CONTIG = (True, False, False)        # contig.rs:300-304  (proper_pairs_only=false)
FF_SEP_STREAM = (True, True, True)      # genome.rs:957-961
FF_SEP_PILEUP = (True, False, False)    # genome.rs:996-1000
FF_NAMES = (True, False, False)         # genome.rs:1032-1036

ZERO7 = ("{s}\tgenome1~random_sequence_length_11000\t0\n{s}\tgenome1~random_sequence_length_11010\t0\n"
         "{s}\tgenome2~seq1\t1.2\n{s}\tgenome3~random_sequence_length_11001\t0\n"
         "{s}\tgenome4~random_sequence_length_11002\t0\n{s}\tgenome5~seq2\t1.2\n"
         "{s}\tgenome6~random_sequence_length_11003\t0\n").format(s=S7)

API_CASES = [
    # ---------------------------------------------------------------- contig.rs
    dict(id="contig_two_contigs_no_zeros", cite="src/contig.rs:325-333", api="contig", bams=[S7 + ".bam"],
         taker="stream", ff=FF_CONTIG, print_zero=False, est=[("mean", 0.0, 0, False)],
         expected=S7 + "\tgenome2~seq1\t1.2\n" + S7 + "\tgenome5~seq2\t1.2\n"),
    dict(id="contig_two_contigs_zeros", cite="src/contig.rs:336-344", api="contig", bams=[S7 + ".bam"],
         taker="stream", ff=FF_CONTIG, print_zero=True, est=[("mean", 0.0, 0, False)], expected=ZERO7),
    dict(id="contig_one_contig_variance", cite="src/contig.rs:379-388", api="contig",
         bams=["2seqs.reads_for_seq1.bam"], taker="stream", ff=FF_CONTIG, print_zero=True,
         est=[("variance", 0.0, 0)],
         expected="2seqs.reads_for_seq1\tseq1\t0.9489489\n2seqs.reads_for_seq1\tseq2\t0\n"),
    dict(id="contig_multiple_methods", cite="src/contig.rs:418-430", api="contig",
         bams=["2seqs.reads_for_seq1.bam"], taker="stream", ff=FF_CONTIG, print_zero=True,
         est=[("mean", 0.0, 0, False), ("variance", 0.0, 0)],
         expected="2seqs.reads_for_seq1\tseq1\t1.2\t0.9489489\n2seqs.reads_for_seq1\tseq2\t0\t0\n"),
    dict(id="contig_julian_error", cite="src/contig.rs:433-445", api="contig",
         bams=["2seqs.reads_for_seq1.with_unmapped.bam"], taker="stream", ff=FF_CONTIG, print_zero=True,
         est=[("mean", 0.0, 0, True)],
         expected="2seqs.reads_for_seq1.with_unmapped\tseq1\t1.497\n2seqs.reads_for_seq1.with_unmapped\tseq2\t1.5\n"),
    dict(id="contig_trimmed_mean_bug", cite="src/contig.rs:448-458", api="contig",
         bams=["2seqs.reads_for_seq1.bam"], taker="stream", ff=FF_CONTIG, print_zero=True,
         est=[("trimmed_mean", 0.0, 0.05, 0.0, 0)],
         expected="2seqs.reads_for_seq1\tseq1\t0\n2seqs.reads_for_seq1\tseq2\t0\n"),
    dict(id="contig_one_zero_no_print_zeroes", cite="src/contig.rs:461-474", api="contig",
         bams=["2seqs.reads_for_seq1.bam"], taker="stream", ff=FF_CONTIG, print_zero=False,
         est=[("mean", 0.0, 0, False), ("trimmed_mean", 0.0, 0.05, 0.0, 0)],
         expected="2seqs.reads_for_seq1\tseq1\t1.2\t0\n"),
    dict(id="contig_one_zero_no_print_zeroes_rev", cite="src/contig.rs:477-490", api="contig",
         bams=["2seqs.reads_for_seq1.bam"], taker="stream", ff=FF_CONTIG, print_zero=False,
         est=[("trimmed_mean", 0.0, 0.05, 0.0, 0), ("mean", 0.0, 0, False)],
         expected="2seqs.reads_for_seq1\tseq1\t0\t1.2\n"),
    dict(id="contig_end_exclusion", cite="src/contig.rs:493-510", api="contig", bams=[S7 + ".bam"],
         taker="stream", ff=FF_CONTIG, print_zero=False, est=[("mean", 0.0, 75, False), ("variance", 0.0, 75)],
         expected=S7 + "\tgenome2~seq1\t1.4117647\t1.3049262\n" + S7 + "\tgenome5~seq2\t1.2435294\t0.6862065\n"),
    dict(id="contig_one_read_of_pair_mapped", cite="src/contig.rs:513-522", api="contig",
         bams=["1read_of_pair_mapped.bam"], taker="stream", ff=FF_CONTIG, print_zero=False,
         est=[("mean", 0.0, 75, True)],
         expected="1read_of_pair_mapped\t73.20100900_E1D.16_contig_9606\t0.011293635\n"),
    dict(id="contig_variance_all_bases_covered", cite="src/contig.rs:525-534", api="contig",
         bams=["k141_2005182.bam"], taker="stream", ff=FF_CONTIG, print_zero=False, est=[("variance", 0.0, 75)],
         expected="k141_2005182\tk141_2005182\t5.107387\n"),
    dict(id="contig_reads_counting_sufficient", cite="src/contig.rs:537-556", api="contig",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], taker="stream", ff=FF_CONTIG, print_zero=False,
         est=[("variance", 0.0, 75)],
         expected="2seqs.reads_for_seq1_and_seq2\tseq1\t1.3049262\n2seqs.reads_for_seq1_and_seq2\tseq2\t0.6862065\n",
         reads_mapped=[(24, 24)]),
    dict(id="contig_reads_counting_insufficient", cite="src/contig.rs:559-577", api="contig",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], taker="stream", ff=FF_CONTIG, print_zero=False,
         est=[("variance", 0.99, 75)], expected="", reads_mapped=[(0, 24)]),

    # ---------------------------------------------------------------- genome.rs, separator / single-genome
    dict(id="sep_first_covered", cite="src/genome.rs:1088-1099", api="sep", bams=["2seqs.reads_for_seq1.bam"],
         sep="q", single=False, taker="stream", ff=FF_SEP_STREAM, print_zero=True, est=[("mean", 0.0, 0, False)],
         expected="2seqs.reads_for_seq1\tse\t0.6\n"),
    dict(id="sep_second_covered", cite="src/genome.rs:1117-1128", api="sep", bams=["2seqs.reads_for_seq2.bam"],
         sep="q", single=False, taker="stream", ff=FF_SEP_STREAM, print_zero=True, est=[("mean", 0.0, 0, False)],
         expected="2seqs.reads_for_seq2\tse\t0.6\n"),
    dict(id="sep_both_covered", cite="src/genome.rs:1146-1159", api="sep",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], sep="e", single=False, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("mean", 0.0, 0, False)], expected="2seqs.reads_for_seq1_and_seq2\ts\t1.2\n"),
    dict(id="sep_min_fraction_under", cite="src/genome.rs:1179-1192", api="sep",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], sep="e", single=False, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("mean", 0.76, 0, False)], expected="2seqs.reads_for_seq1_and_seq2\ts\t0\n"),
    dict(id="sep_min_fraction_just_ok", cite="src/genome.rs:1212-1225", api="sep",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], sep="e", single=False, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("mean", 0.759, 0, False)], expected="2seqs.reads_for_seq1_and_seq2\ts\t1.2\n"),
    dict(id="sep_trimmed_mean", cite="src/genome.rs:1245-1260", api="sep",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], sep="e", single=False, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("trimmed_mean", 0.1, 0.9, 0.759, 0)],
         expected="2seqs.reads_for_seq1_and_seq2\ts\t1.08875\n"),
    dict(id="sep_pileup_counts", cite="src/genome.rs:1282-1292", api="sep",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], sep="e", single=False, taker="pileup", ff=FF_SEP_PILEUP,
         print_zero=True, est=[("pileup_counts", 0.0, 0)],
         expected="".join("2seqs.reads_for_seq1_and_seq2\ts\t%d\t%d\n" % (i, n)
                          for i, n in enumerate([482, 922, 371, 164, 61]))),
    dict(id="sep_zero_coverage_genomes", cite="src/genome.rs:1309-1318", api="sep", bams=[S7 + ".bam"], sep="~",
         single=False, taker="stream", ff=FF_SEP_STREAM, print_zero=True, est=[("mean", 0.1, 0, False)],
         expected="".join("%s\tgenome%d\t%s\n" % (S7, g, c) for g, c in
                          [(1, "0"), (2, "1.2"), (3, "0"), (4, "0"), (5, "1.2"), (6, "0")])),
    dict(id="sep_zero_coverage_genomes_nozeros", cite="src/genome.rs:1320-1327", api="sep", bams=[S7 + ".bam"],
         sep="~", single=False, taker="stream", ff=FF_SEP_STREAM, print_zero=False, est=[("mean", 0.1, 0, False)],
         expected=S7 + "\tgenome2\t1.2\n" + S7 + "\tgenome5\t1.2\n"),
    dict(id="sep_zero_after_min_fraction", cite="src/genome.rs:1373-1383", api="sep", bams=[S7 + ".bam"], sep="~",
         single=False, taker="stream", ff=FF_SEP_STREAM, print_zero=True, est=[("mean", 0.759, 0, False)],
         expected="".join("%s\tgenome%d\t%s\n" % (S7, g, c) for g, c in
                          [(1, "0"), (2, "0"), (3, "0"), (4, "0"), (5, "1.2"), (6, "0")])),
    dict(id="sep_single_genome", cite="src/genome.rs:1385-1398", api="sep", bams=[S7 + ".bam"], sep="~",
         single=True, taker="stream", ff=FF_SEP_STREAM, print_zero=True, est=[("mean", 0.0, 0, False)],
         expected=S7 + "\tgenome1\t0.04209345\n"),
    dict(id="sep_covered_bases", cite="src/genome.rs:1400-1417", api="sep", bams=[S7 + ".bam"], sep="~",
         single=False, taker="stream", ff=FF_SEP_STREAM, print_zero=False, est=[("covered_bases", 0.0)],
         expected=S7 + "\tgenome2\t669\n" + S7 + "\tgenome5\t849\n"),
    dict(id="sep_julian_error", cite="src/genome.rs:1582-1603", api="sep",
         bams=["2seqs.reads_for_seq1.with_unmapped.bam"], sep="\0", single=True, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("mean", 0.1, 0, True)],
         expected="2seqs.reads_for_seq1.with_unmapped\tgenome1\t1.4985\n", reads_mapped=[(20, 24)]),
    dict(id="sep_one_zero_single_genome", cite="src/genome.rs:1605-1621", api="sep",
         bams=["2seqs.reads_for_seq1.bam"], sep="q", single=True, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("mean", 0.0, 0, False), ("trimmed_mean", 0.0, 0.05, 0.0, 0)],
         expected="2seqs.reads_for_seq1\tgenome1\t0.6\t0\n"),
    dict(id="sep_one_zero_single_genome_rev", cite="src/genome.rs:1623-1639", api="sep",
         bams=["2seqs.reads_for_seq1.bam"], sep="q", single=True, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("trimmed_mean", 0.0, 0.05, 0.0, 0), ("mean", 0.0, 0, False)],
         expected="2seqs.reads_for_seq1\tgenome1\t0\t0.6\n"),
    dict(id="sep_one_zero_separator", cite="src/genome.rs:1641-1657", api="sep", bams=["7seqs.reads_for_seq1.bam"],
         sep="~", single=False, taker="stream", ff=FF_SEP_STREAM, print_zero=True,
         est=[("mean", 0.0, 0, False), ("trimmed_mean", 0.0, 0.05, 0.0, 0)],
         expected="".join("7seqs.reads_for_seq1\tgenome%d\t%s\t0\n" % (g, c) for g, c in
                          [(1, "0"), (2, "1.2"), (3, "0"), (4, "0"), (5, "0"), (6, "0")])),
    dict(id="sep_one_zero_separator_rev", cite="src/genome.rs:1659-1675", api="sep",
         bams=["7seqs.reads_for_seq1.bam"], sep="~", single=False, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("trimmed_mean", 0.0, 0.05, 0.0, 0), ("mean", 0.0, 0, False)],
         expected="".join("7seqs.reads_for_seq1\tgenome%d\t0\t%s\n" % (g, c) for g, c in
                          [(1, "0"), (2, "1.2"), (3, "0"), (4, "0"), (5, "0"), (6, "0")])),
    dict(id="sep_read_count", cite="src/genome.rs:1794-1836", api="sep",
         bams=["7seqs.reads_for_seq1.bam", S7 + ".bam"], sep="~", single=False, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("read_count",)],
         expected="".join("7seqs.reads_for_seq1\tgenome%d\t%s\n" % (g, c) for g, c in
                          [(1, "0"), (2, "12"), (3, "0"), (4, "0"), (5, "0"), (6, "0")]) +
                  "".join("%s\tgenome%d\t%s\n" % (S7, g, c) for g, c in
                          [(1, "0"), (2, "12"), (3, "0"), (4, "0"), (5, "12"), (6, "0")]),
         reads_mapped=[(12, 12), (24, 24)]),
    dict(id="sep_read_count_and_fraction", cite="src/genome.rs:1838-1881", api="sep",
         bams=["7seqs.reads_for_seq1.bam", S7 + ".bam"], sep="~", single=False, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("read_count",), ("covered_fraction", 0.1)],
         expected="".join("7seqs.reads_for_seq1\tgenome%d\t%s\n" % (g, c) for g, c in
                          [(1, "0\t0"), (2, "12\t0.727"), (3, "0\t0"), (4, "0\t0"), (5, "0\t0"), (6, "0\t0")]) +
                  "".join("%s\tgenome%d\t%s\n" % (S7, g, c) for g, c in
                          [(1, "0\t0"), (2, "12\t0.669"), (3, "0\t0"), (4, "0\t0"), (5, "12\t0.849"), (6, "0\t0")]),
         reads_mapped=[(12, 12), (24, 24)]),
    dict(id="sep_fraction_fails", cite="src/genome.rs:1883-1922", api="sep",
         bams=["7seqs.reads_for_seq1.bam", S7 + ".bam"], sep="~", single=False, taker="stream", ff=FF_SEP_STREAM,
         print_zero=True, est=[("covered_fraction", 0.99)],
         expected="".join("7seqs.reads_for_seq1\tgenome%d\t0\n" % g for g in range(1, 7)) +
                  "".join("%s\tgenome%d\t0\n" % (S7, g) for g in range(1, 7)),
         reads_mapped=[(0, 12), (0, 24)]),

    # ---------------------------------------------------------------- genome.rs, contig-names mode
    dict(id="names_first_covered", cite="src/genome.rs:1101-1115", api="names", bams=["2seqs.reads_for_seq1.bam"],
         geco=GECO_SE, taker="stream", ff=FF_NAMES, print_zero=True, est=[("mean", 0.0, 0, False)],
         expected="2seqs.reads_for_seq1\tse\t0.6\n"),
    dict(id="names_second_covered", cite="src/genome.rs:1130-1144", api="names", bams=["2seqs.reads_for_seq2.bam"],
         geco=GECO_SE, taker="stream", ff=FF_NAMES, print_zero=True, est=[("mean", 0.0, 0, False)],
         expected="2seqs.reads_for_seq2\tse\t0.6\n"),
    dict(id="names_both_covered", cite="src/genome.rs:1161-1177", api="names",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], geco=GECO_S, taker="stream", ff=FF_NAMES, print_zero=True,
         est=[("mean", 0.0, 0, False)], expected="2seqs.reads_for_seq1_and_seq2\ts\t1.2\n"),
    dict(id="names_min_fraction_under", cite="src/genome.rs:1194-1210", api="names",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], geco=GECO_S, taker="stream", ff=FF_NAMES, print_zero=False,
         est=[("mean", 0.76, 0, False)], expected=""),
    dict(id="names_min_fraction_just_ok", cite="src/genome.rs:1227-1243", api="names",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], geco=GECO_S, taker="stream", ff=FF_NAMES, print_zero=True,
         est=[("mean", 0.759, 0, False)], expected="2seqs.reads_for_seq1_and_seq2\ts\t1.2\n"),
    dict(id="names_trimmed_mean", cite="src/genome.rs:1262-1280", api="names",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], geco=GECO_S, taker="stream", ff=FF_NAMES, print_zero=True,
         est=[("trimmed_mean", 0.1, 0.9, 0.0, 0)], expected="2seqs.reads_for_seq1_and_seq2\ts\t1.08875\n"),
    dict(id="names_pileup_counts", cite="src/genome.rs:1294-1307", api="names",
         bams=["2seqs.reads_for_seq1_and_seq2.bam"], geco=GECO_S, taker="pileup", ff=FF_NAMES, print_zero=True,
         est=[("pileup_counts", 0.0, 0)],
         expected="".join("2seqs.reads_for_seq1_and_seq2\ts\t%d\t%d\n" % (i, n)
                          for i, n in enumerate([482, 922, 371, 164, 61]))),
    dict(id="names_zero_coverage_genomes", cite="src/genome.rs:1438-1475", api="names", bams=[S7 + ".bam"],
         geco=GECO_7, taker="stream", ff=FF_NAMES, print_zero=True, est=[("mean", 0.1, 0, False)],
         expected="".join("%s\tgenome%d\t%s\n" % (S7, g, c) for g, c in
                          [(1, "0"), (2, "1.2"), (3, "0"), (4, "0"), (5, "1.2"), (6, "0")])),
    dict(id="names_zero_coverage_genomes_nozeros", cite="src/genome.rs:1477-1483", api="names", bams=[S7 + ".bam"],
         geco=GECO_7, taker="stream", ff=FF_NAMES, print_zero=False, est=[("mean", 0.1, 0, False)],
         expected=S7 + "\tgenome2\t1.2\n" + S7 + "\tgenome5\t1.2\n"),
    dict(id="names_multiple_methods", cite="src/genome.rs:1486-1526", api="names", bams=[S7 + ".bam"], geco=GECO_7,
         taker="stream", ff=FF_NAMES, print_zero=True, est=[("mean", 0.1, 0, False), ("variance", 0.1, 0)],
         expected="".join("%s\tgenome%d\t%s\n" % (S7, g, c) for g, c in
                          [(1, "0\t0"), (2, "1.2\t1.3633634"), (3, "0\t0"), (4, "0\t0"), (5, "1.2\t0.6166166"),
                           (6, "0\t0")])),
    dict(id="names_multiple_methods_nozeros", cite="src/genome.rs:1528-1542", api="names", bams=[S7 + ".bam"],
         geco=GECO_7, taker="stream", ff=FF_NAMES, print_zero=False,
         est=[("mean", 0.1, 0, False), ("variance", 0.1, 0)],
         expected=S7 + "\tgenome2\t1.2\t1.3633634\n" + S7 + "\tgenome5\t1.2\t0.6166166\n", reads_mapped=[(24, 24)]),
    dict(id="names_reads_mapped_subset", cite="src/genome.rs:1545-1580", api="names", bams=[S7 + ".bam"],
         geco=GECO_23, taker="stream", ff=FF_NAMES, print_zero=True,
         est=[("mean", 0.1, 0, False), ("variance", 0.1, 0)],
         expected=S7 + "\tgenome2\t1.2\t1.3633634\n" + S7 + "\tgenome3\t0\t0\n", reads_mapped=[(12, 24)]),
    dict(id="names_below_min_covered", cite="src/genome.rs:1965-1985", api="names", bams=[S7 + ".bam"],
         geco=GECO_7, taker="stream", ff=FF_NAMES, print_zero=False,
         est=[("mean", 0.99, 0, False), ("variance", 0.99, 0)], expected="", reads_mapped=[(0, 24)]),
]

# ReferenceSortedBamFilter::new(reader, flags, min_aligned_length_single, min_percent_identity_single,
#   min_aligned_percent_single, min_mapq_single, min_aligned_length_pair, min_percent_identity_pair,
#   min_aligned_percent_pair, filter_out=true) -> expected qname order (filter.rs tests).
FILTER_CASES = [
    dict(id="filter_hello_world", cite="src/filter.rs:342-373", bam=S7 + ".bam", ff=(False, False, False),
         single=(0, 0.0, 0.0), mapq=0, pair=(90, 0.99, 0.0), mode=(False, True),
         qnames="9 9 12 12 7 7 11 11 10 10 8 8 4 4 6 6 1 1 2 2 3 3 5 5".split(), exhaustive=True),
    dict(id="filter_one_bad_read_a", cite="src/filter.rs:406-430", bam="2seqs.bad_read.1.bam",
         ff=(False, False, False), single=(0, 0.0, 0.0), mapq=0, pair=(250, 0.99, 0.0), mode=None,
         qnames="2 2 3 3".split(), exhaustive=False),
    dict(id="filter_one_bad_read_b", cite="src/filter.rs:432-454", bam="2seqs.bad_read.1.bam",
         ff=(False, False, False), single=(0, 0.0, 0.0), mapq=0, pair=(300, 0.98, 0.0), mode=None,
         qnames="2 2 3 3".split(), exhaustive=False),
    dict(id="filter_one_bad_read_c", cite="src/filter.rs:456-478", bam="2seqs.bad_read.1.with_extra.bam",
         ff=(False, False, False), single=(0, 0.0, 0.0), mapq=0, pair=(0, 0.98, 0.94), mode=None,
         qnames="2 2 3 3".split(), exhaustive=False),
    dict(id="filter_one_bad_read_d", cite="src/filter.rs:480-502", bam="2seqs.bad_read.1.bam",
         ff=(False, False, False), single=(0, 0.0, 0.0), mapq=0, pair=(299, 0.98, 0.0), mode=None,
         qnames="1 1 2 2".split(), exhaustive=False),
    dict(id="filter_single_reads", cite="src/filter.rs:605-633", bam="2seqs.bad_read.1.bam",
         ff=(True, False, False), single=(0, 0.99, 0.0), mapq=0, pair=(0, 0.0, 0.0), mode=(True, False),
         qnames="2 3 4 1".split(), exhaustive=False),
    dict(id="filter_single_and_paired", cite="src/filter.rs:665-693", bam="2seqs.bad_read.1.bam",
         ff=(False, False, False), single=(0, 0.95, 0.0), mapq=0, pair=(300, 0.0, 0.0), mode=(True, True),
         qnames="2 2 3 3 4 4".split(), exhaustive=False),
    dict(id="filter_negative_insert", cite="src/filter.rs:725-754", bam="eg2.bam", ff=(False, False, False),
         single=(0, 0.0, 0.0), mapq=0, pair=(1, 0.0, 0.0), mode=(False, True), count=11192),
    dict(id="filter_mapq_no_bads", cite="src/filter.rs:756-784", bam="mapq_test.sam", ff=(True, False, False),
         single=(0, 0.0, 0.0), mapq=1, pair=(0, 0.0, 0.0), mode=(True, False), qnames="1 1 2 2".split(),
         exhaustive=False),
    dict(id="filter_mapq_single_bad", cite="src/filter.rs:786-814", bam="mapq_test.sam", ff=(True, False, False),
         single=(0, 0.0, 0.0), mapq=51, pair=(0, 0.0, 0.0), mode=(True, False), qnames="1 2 2".split(),
         exhaustive=False),
    dict(id="filter_mapq_pairs_one_bad", cite="src/filter.rs:816-844", bam="mapq_test.sam", ff=(True, False, False),
         single=(0, 0.0, 0.0), mapq=51, pair=(1, 0.0, 0.0), mode=(False, True), qnames="2 2".split(),
         exhaustive=False),
]
# the same constructor with filter_out = false (`coverm filter --inverse`): expected qname order
FILTER_INVERSE_CASES = [
    dict(id="filter_hello_world_inverse", cite="src/filter.rs:375-404", bam=S7 + ".bam", ff=(False, False, False),
         single=(0, 0.0, 0.0), mapq=0, pair=(90, 0.99, 0.0), qnames=[], exhaustive=True),
    dict(id="filter_one_bad_read_inverse_a", cite="src/filter.rs:505-529", bam="2seqs.bad_read.1.bam", ff=(False, False, False),
         single=(0, 0.0, 0.0), mapq=0, pair=(250, 0.99, 0.0), qnames="1 1".split(), exhaustive=False),
    dict(id="filter_one_bad_read_inverse_b", cite="src/filter.rs:531-553", bam="2seqs.bad_read.1.bam", ff=(False, False, False),
         single=(0, 0.0, 0.0), mapq=0, pair=(300, 0.98, 0.0), qnames="1 1".split(), exhaustive=False),
    dict(id="filter_one_bad_read_inverse_c", cite="src/filter.rs:555-577", bam="2seqs.bad_read.1.with_extra.bam", ff=(False, False, False),
         single=(0, 0.0, 0.0), mapq=0, pair=(0, 0.98, 0.94), qnames="1 1".split(), exhaustive=False),
    dict(id="filter_single_reads_inverse", cite="src/filter.rs:635-662", bam="2seqs.bad_read.1.bam", ff=(True, False, False),
         single=(0, 0.99, 0.0), mapq=0, pair=(0, 0.0, 0.0), qnames="1".split(), exhaustive=False),
    dict(id="filter_single_and_paired_inverse", cite="src/filter.rs:695-722", bam="2seqs.bad_read.1.bam", ff=(False, False, False),
         single=(0, 0.95, 0.0), mapq=0, pair=(300, 0.0, 0.0), qnames="1 1".split(), exhaustive=False),
]
# NB: filter.rs passes min_mapq_single=0 in most cases above; 0 != 255 so MAPQ "filtering" is on with
# threshold 0 (everything >= 0 passes unless mapq == 255) and it participates in mode selection (:48-61).

G7 = ["genome1~random_sequence_length_11000", "genome1~random_sequence_length_11010", "genome2~seq1",
      "genome3~random_sequence_length_11001", "genome4~random_sequence_length_11002", "genome5~seq2",
      "genome6~random_sequence_length_11003"]


def _rows(prefix, names, vals):
    return "".join("%s%s\t%s\n" % (prefix, n, v) for n, v in zip(names, vals))


# `coverm <mode> --bam-files ...` end-to-end expectations (tests/test_cmdline.rs).  args are the
# keyword arguments of the CLI restatement (oracle.run_cli / tests/harness_cli.run).
CLI_CASES = [
    dict(id="cli_relative_abundance_and_mean", cite="tests/test_cmdline.rs:1145-1172", mode="genome",
         bams=[S7 + ".bam"], args=dict(methods=["relative_abundance", "mean"], output_format="sparse", separator="~"),
         match="contains",
         expected="Sample\tGenome\tRelative Abundance (%)\tMean\n" + _rows(S7 + "\t", ["unmapped"] + [
             "genome%d" % g for g in range(1, 7)], ["0\tNA", "0\t0", "53.16792\t1.4117647", "0\t0", "0\t0",
                                                    "46.832077\t1.2435294", "0\t0"])),
    dict(id="cli_contig_dense_simple", cite="tests/test_cmdline.rs:1175-1197", mode="contig", bams=[S7 + ".bam"],
         args=dict(output_format="dense"), match="contains",
         expected="Contig\t" + S7 + " Mean\n" + _rows("", G7, ["0", "0", "1.4117647", "0", "0", "1.2435294", "0"])),
    dict(id="cli_genome_dense_simple", cite="tests/test_cmdline.rs:1200-1226", mode="genome", bams=[S7 + ".bam"],
         args=dict(methods=["relative_abundance"], separator="~", output_format="dense"), match="contains",
         expected="Genome\t" + S7 + " Relative Abundance (%)\n" + _rows("", ["unmapped"] + [
             "genome%d" % g for g in range(1, 7)], ["0", "0", "53.167923", "0", "0", "46.832077", "0"])),
    dict(id="cli_metabat_supplementary", cite="tests/test_cmdline.rs:1562-1578", mode="contig",
         bams=["k141_7.reheadered.bam"], args=dict(methods=["metabat"]), match="contains",
         expected="contigName\tcontigLen\ttotalAvgDepth\tk141_7.reheadered.bam\tk141_7.reheadered.bam-var\n"
                  "k141_7\t350\t0.69\t0.69\t2.0843"),
    dict(id="cli_metabat_97_of_100", cite="tests/test_cmdline.rs:1581-1598", mode="contig",
         bams=["k141_2005182.head11.bam"], args=dict(methods=["metabat"]), match="contains",
         expected="contigName\tcontigLen\ttotalAvgDepth\tk141_2005182.head11.bam\tk141_2005182.head11.bam-var\n"
                  "k141_2005182\t225\t1.9333\t1.9333\t0.0631"),
    dict(id="cli_metabat_deletions", cite="tests/test_cmdline.rs:1601-1612", mode="contig",
         bams=["k141_109815.stray_read.bam"], args=dict(methods=["metabat"]), match="contains",
         expected="contigName\tcontigLen\ttotalAvgDepth\tk141_109815.stray_read.bam\t"
                  "k141_109815.stray_read.bam-var\nk141_109815\t362\t0.6274\t0.6274\t0.2349"),
    dict(id="cli_genome_definition", cite="tests/test_cmdline.rs:2263-2280", mode="genome", bams=[S7 + ".bam"],
         args=dict(genome_definition="7seqs.definition"), match="contains_all",
         expected=["Genome\t" + S7 + " Relative Abundance (%)\n", "genome2\t53.167923\n", "genome5\t46.832077\n"]),
    dict(id="cli_contig_sparse_rpkm", cite="tests/test_cmdline.rs:2466-2491", mode="contig",
         bams=["7seqs.fnaVbad_read.bam"],
         args=dict(methods=["rpkm", "reads_per_base", "length", "count"], output_format="sparse"), match="is",
         expected="Sample\tContig\tRPKM\tReads per base\tLength\tRead Count\n" + _rows("7seqs.fnaVbad_read\t", G7, [
             "0\t0\t11000\t0", "0\t0\t11010\t0", "500000\t0.01\t1000\t10", "0\t0\t11001\t0", "0\t0\t11002\t0",
             "500000\t0.01\t1000\t10", "0\t0\t11003\t0"])),
    dict(id="cli_contig_dense_rpkm", cite="tests/test_cmdline.rs:2494-2517", mode="contig",
         bams=["7seqs.fnaVbad_read.bam"], args=dict(methods=["rpkm", "reads_per_base", "length", "count"]),
         match="is",
         expected="Contig\t7seqs.fnaVbad_read RPKM\t7seqs.fnaVbad_read Reads per base\t7seqs.fnaVbad_read Length"
                  "\t7seqs.fnaVbad_read Read Count\n" + _rows("", G7, [
             "0\t0\t11000\t0", "0\t0\t11010\t0", "500000\t0.01\t1000\t10", "0\t0\t11001\t0", "0\t0\t11002\t0",
             "500000\t0.01\t1000\t10", "0\t0\t11003\t0"])),
    dict(id="cli_single_genome_dense_rpkm", cite="tests/test_cmdline.rs:2520-2540", mode="genome",
         bams=["7seqs.fnaVbad_read.bam"],
         args=dict(single_genome=True, methods=["rpkm", "reads_per_base", "length", "count"],
                   min_covered_fraction=0), match="is",
         expected="Genome\t7seqs.fnaVbad_read RPKM\t7seqs.fnaVbad_read Reads per base\t7seqs.fnaVbad_read Length"
                  "\t7seqs.fnaVbad_read Read Count\ngenome1\t17538.936\t0.00035077872\t57016\t20\n"),
    dict(id="cli_single_genome_rpkm_min_covered", cite="tests/test_cmdline.rs:2543-2558", mode="genome",
         bams=["7seqs.fnaVbad_read.bam"], args=dict(single_genome=True, methods=["rpkm"]), match="is",
         expected="Genome\t7seqs.fnaVbad_read RPKM\ngenome1\t0\n"),
    dict(id="cli_genome_all_methods", cite="tests/test_cmdline.rs:2785-2814", mode="genome",
         bams=["7seqs.fnaVbad_read.bam"],
         # --genome-fasta-directory tests/data/genomes_dir_7seqs == one genome per FASTA named genomeN, holding
         # the contigs listed in 7seqs.definition; the table comparison sorts rows (assert_equal_table :17-31).
         args=dict(output_format="sparse", genome_definition="7seqs.definition",
                   methods=["covered_bases", "covered_fraction", "mean", "variance", "trimmed_mean", "rpkm",
                            "relative_abundance", "length"], min_covered_fraction=0), match="table",
         expected="Sample\tGenome\tCovered Bases\tCovered Fraction\tMean\tVariance\tTrimmed Mean\tRPKM\t"
                  "Relative Abundance (%)\tLength\n"
                  "7seqs.fnaVbad_read\tunmapped\tNA\tNA\tNA\tNA\tNA\tNA\t0\tNA\n"
                  "7seqs.fnaVbad_read\tgenome2\t899\t0.899\t1.6764706\t0.51357985\t1.6788511\t500000\t50\t1000\n"
                  "7seqs.fnaVbad_read\tgenome6\t0\t0\t0\t0\t0\t0\t0\t11003\n"
                  "7seqs.fnaVbad_read\tgenome4\t0\t0\t0\t0\t0\t0\t0\t11002\n"
                  "7seqs.fnaVbad_read\tgenome3\t0\t0\t0\t0\t0\t0\t0\t11001\n"
                  "7seqs.fnaVbad_read\tgenome5\t900\t0.9\t1.6764706\t0.51357985\t1.6788511\t500000\t50\t1000\n"
                  "7seqs.fnaVbad_read\tgenome1\t0\t0\t0\t0\t0\t0\t0\t22010\n"),
    dict(id="cli_contig_unsorted", cite="tests/test_cmdline.rs:3073-3080", mode="contig",
         bams=["2seqs.bad_read.1.unsorted.bam"], args=dict(), match="error",
         expected="BAM file appears to be unsorted"),
    dict(id="cli_genome_sep_unsorted", cite="tests/test_cmdline.rs:3083-3096", mode="genome",
         bams=["2seqs.bad_read.1.unsorted.bam"], args=dict(separator="e"), match="error",
         expected="BAM file appears to be unsorted"),
    dict(id="cli_genome_names_unsorted", cite="tests/test_cmdline.rs:3098-3114 (genomes_dir given as the equivalent definition)",
         mode="genome", bams=["2seqs.bad_read.1.unsorted.bam"], args=dict(genome_definition="2seqs.by-file.definition"),
         match="error", expected="BAM file appears to be unsorted"),
    dict(id="cli_tpm_contig_sparse", cite="tests/test_cmdline.rs:3457-3480", mode="contig", bams=["tpm_test.bam"],
         args=dict(output_format="sparse", methods=["mean", "tpm"]), match="is",
         expected="Sample\tContig\tMean\tTPM\n" + _rows("tpm_test\t", G7, [
             "0\t0", "0\t0", "1.5882353\t900000.0357627869", "0\t0", "0\t0", "0.14467005\t99999.99403953552",
             "0\t0"])),
    dict(id="cli_tpm_contig_dense", cite="tests/test_cmdline.rs:3483-3504", mode="contig", bams=["tpm_test.bam"],
         args=dict(methods=["mean", "tpm"]), match="is",
         expected="Contig\ttpm_test Mean\ttpm_test TPM\n" + _rows("", G7, [
             "0\t0", "0\t0", "1.5882353\t900000.06", "0\t0", "0\t0", "0.14467005\t99999.99", "0\t0"])),
    dict(id="cli_tpm_genome_sparse", cite="tests/test_cmdline.rs:3507-3533", mode="genome", bams=["tpm_test.bam"],
         args=dict(output_format="sparse", methods=["mean", "tpm"], separator="~", min_covered_fraction=0),
         match="is",
         expected="Sample\tGenome\tMean\tTPM\n" + _rows("tpm_test\t", ["genome%d" % g for g in range(1, 7)], [
             "0\t0", "1.5882353\t900000.0357627869", "0\t0", "0\t0", "0.14467005\t99999.99403953552", "0\t0"])),
    dict(id="cli_tpm_genome_dense", cite="tests/test_cmdline.rs:3536-3560", mode="genome", bams=["tpm_test.bam"],
         args=dict(methods=["mean", "tpm"], separator="~", min_covered_fraction=0), match="is",
         expected="Genome\ttpm_test Mean\ttpm_test TPM\n" + _rows("", ["genome%d" % g for g in range(1, 7)], [
             "0\t0", "1.5882353\t900000.06", "0\t0", "0\t0", "0.14467005\t99999.99", "0\t0"])),
    dict(id="cli_single_genome_supplementary_count", cite="tests/test_cmdline.rs:3586-3604", mode="genome",
         bams=["2seqs.bad_read.1.with_supplementary.bam"],
         args=dict(methods=["count"], single_genome=True, min_covered_fraction=0), match="is",
         expected="Genome\t2seqs.bad_read.1.with_supplementary Read Count\ngenome1\t20\n"),
    dict(id="cli_mapq_none", cite="tests/test_cmdline.rs:4070-4090", mode="genome", bams=["mapq_test.sam"],
         args=dict(methods=["mean", "covered_fraction"], single_genome=True, min_covered_fraction=0), match="is",
         expected="Genome\tmapq_test Mean\tmapq_test Covered Fraction\ngenome1\t0.009380695\t0.00875193\n"),
    dict(id="cli_mapq_all_out", cite="tests/test_cmdline.rs:4092-4110", mode="genome", bams=["mapq_test.sam"],
         args=dict(methods=["mean", "covered_fraction"], single_genome=True, min_covered_fraction=0, min_mapq=100),
         match="is", expected="Genome\tmapq_test Mean\tmapq_test Covered Fraction\ngenome1\t0\t0\n"),
    dict(id="cli_mapq_contig_none", cite="tests/test_cmdline.rs:4114-4137", mode="contig", bams=["mapq_test.sam"],
         args=dict(methods=["mean", "covered_fraction"]), match="is",
         expected="Contig\tmapq_test Mean\tmapq_test Covered Fraction\n" + _rows("", G7, [
             "0\t0", "0\t0", "0.61764705\t0.499", "0\t0", "0\t0", "0\t0", "0\t0"])),
    dict(id="cli_mapq_contig_51_single", cite="tests/test_cmdline.rs:4139-4160", mode="contig",
         bams=["mapq_test.sam"], args=dict(methods=["mean", "covered_fraction"], min_mapq=51), match="is",
         expected="Contig\tmapq_test Mean\tmapq_test Covered Fraction\n" + _rows("", G7, [
             "0\t0", "0\t0", "0.5294118\t0.4", "0\t0", "0\t0", "0\t0", "0\t0"])),
    dict(id="cli_mapq_contig_51_pairs", cite="tests/test_cmdline.rs:4164-4188", mode="contig",
         bams=["mapq_test.sam"], args=dict(methods=["mean", "covered_fraction"], min_mapq=51,
                                           proper_pairs_only=True), match="is",
         expected="Contig\tmapq_test Mean\tmapq_test Covered Fraction\n" + _rows("", G7, [
             "0\t0", "0\t0", "0.3529412\t0.3", "0\t0", "0\t0", "0\t0", "0\t0"])),
    dict(id="cli_single_genome_anir", cite="tests/test_cmdline.rs:4191-4208", mode="genome",
         bams=["2seqs.bad_read.1.with_supplementary.bam"],
         args=dict(methods=["anir"], single_genome=True, min_covered_fraction=0), match="is",
         expected="Genome\t2seqs.bad_read.1.with_supplementary ANIr\ngenome1\t0.999\n"),
]

# ---- CoverageTaker / CoveragePrinter unit tests (coverage_takers.rs:383-760, coverage_printer.rs:562-720): a script of
# trait calls ("S" start_stoit, "E" start_entry, "C" add_single_coverage) and what the structure / iterator / printer yields
_TWO = [("S", "stoit1"), ("E", 0, "contig1"), ("C", 1.1), ("C", 1.2), ("E", 3, "contig2"), ("C", 2.1), ("C", 2.2)]
_MISMATCH = _TWO + [("S", "stoit2"), ("E", 1, "contig1.5"), ("C", 10.1), ("C", 10.2), ("E", 3, "contig2"), ("C", 20.1), ("C", 20.2),
                    ("E", 5, "contig5"), ("C", 20.1), ("C", 20.2)]
_HELLO = [("S", "stoit1"), ("E", 0, "contig1"), ("C", 1.1), ("C", 1.2)]
TAKER_CASES = [
    dict(id="taker_cached_hello_world", cite="coverage_takers.rs:383-419", n=2, script=_HELLO,
         stoits=["stoit1"], entries=["contig1"], coverages=[[(0, 1.1), (0, 1.2)]]),
    dict(id="taker_cached_two_samples_matching", cite="coverage_takers.rs:421-506", n=2,
         script=_TWO + [("S", "stoit2"), ("E", 0, "contig1"), ("C", 10.1), ("C", 10.2), ("E", 3, "contig2"), ("C", 20.1), ("C", 20.2)],
         stoits=["stoit1", "stoit2"], entries=["contig1", None, None, "contig2"],
         coverages=[[(0, 1.1), (0, 1.2), (3, 2.1), (3, 2.2)], [(0, 10.1), (0, 10.2), (3, 20.1), (3, 20.2)]]),
    dict(id="taker_cached_two_samples_mismatching", cite="coverage_takers.rs:508-606", n=2, script=_MISMATCH,
         stoits=["stoit1", "stoit2"], entries=["contig1", "contig1.5", None, "contig2", None, "contig5"],
         coverages=[[(0, 1.1), (0, 1.2), (3, 2.1), (3, 2.2)],
                    [(1, 10.1), (1, 10.2), (3, 20.1), (3, 20.2), (5, 20.1), (5, 20.2)]]),
    dict(id="taker_cached_next", cite="coverage_takers.rs:608-698", n=2, script=_MISMATCH,
         iterate=[(0, 0, [1.1, 1.2]), (1, 0, [0.0, 0.0]), (3, 0, [2.1, 2.2]), (5, 0, [0.0, 0.0]), (0, 1, [0.0, 0.0]),
                  (1, 1, [10.1, 10.2]), (3, 1, [20.1, 20.2]), (5, 1, [20.1, 20.2])]),
    dict(id="taker_cached_next_one_coverage", cite="coverage_takers.rs:700-760", n=1,
         script=[("S", "stoit1"), ("E", 0, "contig1"), ("C", 1.1), ("E", 3, "contig2"), ("C", 2.1), ("S", "stoit2"),
                 ("E", 1, "contig1.5"), ("C", 10.1), ("E", 3, "contig2"), ("C", 20.1), ("E", 5, "contig5"), ("C", 20.1)],
         iterate=[(0, 0, [1.1]), (1, 0, [0.0]), (3, 0, [2.1]), (5, 0, [0.0]), (0, 1, [0.0]), (1, 1, [10.1]), (3, 1, [20.1]),
                  (5, 1, [20.1])]),
    dict(id="printer_dense_hello_world", cite="coverage_printer.rs:562-584", n=2, script=_HELLO, printer="dense",
         headers=["mean", "std"], text="Contig\tstoit1 mean\tstoit1 std\ncontig1\t1.1\t1.2\n"),
    dict(id="printer_dense_newline", cite="coverage_printer.rs:586-609", n=2,
         script=[("S", "stoit1"), ("E", 0, "contig1\r"), ("C", 1.1), ("C", 1.2)], printer="dense", headers=["mean", "std"],
         text="Contig\tstoit1 mean\tstoit1 std\ncontig1\t1.1\t1.2\n"),
    dict(id="printer_dense_easy_normalised", cite="coverage_printer.rs:611-638", n=2, script=_HELLO, printer="dense",
         headers=["mean", "std"], reads_mapped=[(1, 2)], normalise=[0],
         text="Contig\tstoit1 mean\tstoit1 std\nunmapped\t50\tNA\ncontig1\t50\t1.2\n"),
    dict(id="printer_metabat_easy", cite="coverage_printer.rs:640-680", n=3, printer="metabat", headers=[],
         script=[("S", "stoit1"), ("E", 0, "contig1"), ("C", 1024.0), ("C", 1.1), ("C", 1.2), ("E", 1, "contig2"), ("C", 1025.0),
                 ("C", 2.1), ("C", 2.2), ("S", "stoit2"), ("E", 0, "contig1"), ("C", 1024.0), ("C", 21.1), ("C", 21.2),
                 ("E", 1, "contig2"), ("C", 1025.0), ("C", 22.1), ("C", 22.2)],
         text="contigName\tcontigLen\ttotalAvgDepth\tstoit1.bam\tstoit1.bam-var\tstoit2.bam\tstoit2.bam-var\n"
              "contig1\t1024\t11.1\t1.1\t1.2\t21.1\t21.2\ncontig2\t1025\t12.1\t2.1\t2.2\t22.1\t22.2\n"),
    dict(id="printer_sparse_hello_world", cite="coverage_printer.rs:682-695", n=2, script=_HELLO, printer="sparse", headers=[],
         text="stoit1\tcontig1\t1.1\t1.2\n"),
    dict(id="printer_sparse_newline", cite="coverage_printer.rs:697-710", n=2, printer="sparse", headers=[],
         script=[("S", "stoit1"), ("E", 0, "contig1\r"), ("C", 1.1), ("C", 1.2)], text="stoit1\tcontig1\t1.1\t1.2\n"),
]

# ---- per-gene coverage (--gff): genes.rs unit tests :621-766 and tests/test_cmdline.rs:134-206
_G2 = [("gene_seq1", "seq1", 0, 1000), ("gene_seq2", "seq2", 0, 1000)]
_B1 = "2seqs.reads_for_seq1"
GENE_API_CASES = [
    dict(id="genes_whole_contig_matches_contig_coverage", cite="genes.rs:649-684", bam=_B1 + ".bam", genes=_G2,
         est=("mean", 0.0, 0, False), namer=None, print_zeros=True,
         expected=_B1 + "\tgene_seq1\tseq1\t1.2\n" + _B1 + "\tgene_seq2\tseq2\t0\n"),
    dict(id="genes_genome_namer_column", cite="genes.rs:686-723", bam=_B1 + ".bam", genes=_G2,
         est=("mean", 0.0, 0, False), namer={"seq1": "genomeA"}, print_zeros=True,
         expected=_B1 + "\tgene_seq1\tseq1\tgenomeA\t1.2\n"),
    dict(id="genes_no_zeros", cite="genes.rs:725-749", bam=_B1 + ".bam", genes=_G2,
         est=("mean", 0.0, 0, False), namer=None, print_zeros=False, expected=_B1 + "\tgene_seq1\tseq1\t1.2\n"),
    dict(id="genes_count_method", cite="genes.rs:751-765", bam=_B1 + ".bam", genes=_G2[:1],
         est=("count",), namer=None, print_zeros=False, expected=_B1 + "\tgene_seq1\tseq1\t12\n"),
]
GFF_PARSE_EXPECTED = [("gene1", "seq1", 0, 1000), ("gene2", "seq1", 99, 200), ("gene3", "seq2", 0, 1000)]   # genes.rs:621-647
GENE_CLI_CASES = [
    dict(id="cli_contig_per_gene_mean", cite="tests/test_cmdline.rs:134-158", mode="contig", bams=[_B1 + ".bam"],
         args=dict(gff="2seqs.gff", methods=["mean"], contig_end_exclusion=0, output_format="sparse"), match="contains_all",
         expected=["Sample\tGene\tContig\tMean", _B1 + "\tgene1\tseq1\t1.2", _B1 + "\tgene3\tseq2\t0"]),
    dict(id="cli_contig_per_gene_count", cite="tests/test_cmdline.rs:160-179", mode="contig", bams=[_B1 + ".bam"],
         args=dict(gff="2seqs.gff", methods=["count"], output_format="sparse", no_zeros=True), match="contains_all",
         expected=[_B1 + "\tgene1\tseq1\t12"]),
    dict(id="cli_genome_per_gene_mean", cite="tests/test_cmdline.rs:181-206", mode="genome", bams=[_B1 + ".bam"],
         args=dict(gff="2seqs.gff", genome_definition="2seqs.genome-definition", methods=["mean"], contig_end_exclusion=0,
                   min_covered_fraction=0, output_format="sparse"), match="contains_all",
         expected=["Sample\tGene\tContig\tGenome\tMean", _B1 + "\tgene1\tseq1\tgenomeA\t1.2",
                   _B1 + "\tgene3\tseq2\tgenomeB\t0"]),
]

FIXTURE_FILES = sorted({b for c in API_CASES for b in c["bams"]} | {c["bam"] for c in FILTER_CASES}
                       | {b for c in CLI_CASES for b in c["bams"]})
