"""Runs the product orchestrator (coverm_amd/coverm-amd = covh_cli_main) with the keyword arguments oracle.run_cli takes, so that a
parity test reads `binary.run(mode, paths, **args) == O.run_cli(mode, paths, bams=..., **args)`."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "coverm_amd", "coverm-amd")

_FLAGS = {"min_covered_fraction": "--min-covered-fraction", "contig_end_exclusion": "--contig-end-exclusion", "trim_min": "--trim-min",
          "trim_max": "--trim-max", "output_format": "--output-format", "min_read_aligned_length": "--min-read-aligned-length",
          "min_read_percent_identity": "--min-read-percent-identity", "min_read_aligned_percent": "--min-read-aligned-percent",
          "min_mapq": "--min-mapq", "min_read_aligned_length_pair": "--min-read-aligned-length-pair",
          "min_read_percent_identity_pair": "--min-read-percent-identity-pair", "min_read_aligned_percent_pair": "--min-read-aligned-percent-pair",
          "separator": "--separator", "genome_definition": "--genome-definition", "gff": "--gff", "gff_feature_type": "--gff-feature-type"}
_SWITCHES = {"no_zeros": "--no-zeros", "proper_pairs_only": "--proper-pairs-only", "exclude_supplementary": "--exclude-supplementary",
             "include_secondary": "--include-secondary", "single_genome": "--single-genome"}


def argv(mode, paths, threads=8, devices=None, **kw):
    v = [BIN, mode, "-b"] + list(paths)
    if kw.get("methods"):
        v += ["-m"] + list(kw.pop("methods"))
    kw.pop("methods", None)
    for k, val in kw.items():
        if k in _SWITCHES:
            if val:
                v.append(_SWITCHES[k])
        elif k in _FLAGS:
            if val is not None:
                v += [_FLAGS[k], str(val)]
        else:
            raise KeyError("no coverm-amd flag for %r" % k)
    v += ["-t", str(threads)]
    if devices:
        v += ["--devices", devices]
    return v


def run_full(mode, paths, env=None, timeout=900, **kw):
    """The completed process (stdout = the table, stderr = the run's log); raises with stderr when the binary fails."""
    r = subprocess.run(argv(mode, paths, **kw), capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    if r.returncode != 0:
        raise RuntimeError("coverm-amd failed (%d): %s" % (r.returncode, r.stderr[-3000:]))
    return r


def run(mode, paths, env=None, timeout=900, **kw):
    """stdout of the run; raises with stderr when the binary fails."""
    r = subprocess.run(argv(mode, paths, **kw), capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    if r.returncode != 0:
        raise RuntimeError("coverm-amd failed (%d): %s" % (r.returncode, r.stderr[-3000:]))
    return r.stdout
