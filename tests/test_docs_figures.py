"""DESIGN.md's measured figures are generated from the committed evidence under profiles/ (tools/design_figures.py): the block in the
document must be what the generator prints today, so a figure cannot outlive the file that backed it (round 2's DESIGN.md quoted
"95.9 us = 70.0 %" from an earlier refresh of a directory whose final content said 96.97 us)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_md_figures_are_the_generated_ones():
    spec = importlib.util.spec_from_file_location("design_figures", os.path.join(ROOT, "tools", "design_figures.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, b = doc.index(gen.BEGIN), doc.index(gen.END) + len(gen.END)
    assert doc[a:b] == gen.block(), "DESIGN.md is stale: run python3 tools/design_figures.py --write"
    assert len(doc.encode()) < 30 * 1024, "DESIGN.md states what is (eight sections, < 30 KB); the notebook goes to NOTES.md"


def test_manifest_names_existing_files():
    """profiles/MANIFEST.md maps quoted figures to committed files: every path it names must exist."""
    import re
    text = open(os.path.join(ROOT, "profiles", "MANIFEST.md")).read()
    paths = set(re.findall(r"`(profiles/[^`*]+?)`", text))
    assert len(paths) >= 10
    missing = [p for p in paths if not os.path.exists(os.path.join(ROOT, p))]
    assert not missing, missing
