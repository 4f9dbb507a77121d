"""DESIGN.md / README.md quote the round's figures only through a block that tools/check_docs.py generates from the committed files under
profiles/ — this test fails when a block is stale, when a cited file is missing, or when the prose cites a profiles/ path that does not
exist (the files were overwritten after the prose that quoted them)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documents_follow_the_committed_profiles():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_docs.py")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
