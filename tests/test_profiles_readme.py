"""profiles/README.md is GENERATED from the committed measurement files (scripts/make_profiles_readme.py): regenerating it must reproduce the
committed text, i.e. every number quoted there is the one in the artefact it names."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_profiles_readme_is_what_the_generator_writes(tmp_path):
    path = os.path.join(ROOT, "profiles", "README.md")
    before = open(path).read()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_profiles_readme.py")], cwd=ROOT, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        after = open(path).read()
    finally:
        open(path, "w").write(before)
    assert after == before, "profiles/README.md is stale: run python scripts/make_profiles_readme.py"
