"""Stamp the committed counter summaries bench.py quotes (roofline.traffic, roofline.mfma_busy, secondary.roofline.*) with the hash of
the attention kernel sources they were collected on -> profiles/pmc_manifest.json.  Run right after the PMC passes of a tree:

    python tools/pmc_manifest.py <tree tag or commit> <traffic.json> <mfma_busy.txt> <sampler summary.json>

bench.py refuses (prints null + "stale") any of the three whose stamp differs from the sources it is running on."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from osufusion_amd.csrc.build import ATTN_SOURCES, source_hash  # noqa: E402

tree, traffic, busy, sampler = sys.argv[1:5]
out = {"tree": tree, "attn_sources": ATTN_SOURCES, "attn_source_sha256": source_hash(), "traffic": traffic, "mfma_busy": busy, "sampler": sampler}
(ROOT / "profiles" / "pmc_manifest.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
