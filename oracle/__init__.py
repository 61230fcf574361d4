"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (torch fp32 / numpy / plain C) of the reference hot path
(BatsResearch/menghini-neurips23-code: models/clip_encoders.py,
models/prompts_models.py, utils/clip_pseudolabels.py and the third-party
openai `clip` arithmetic they drive).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import anything from here, and only as the
checker -- never as the thing measured or shipped.  The product package
(`menghini-neurips23-code_amd/`, import name `grip_amd`) must never import it.

Pinning: the reference ships no tests or golden vectors for this path and its
arithmetic lives in the un-vendored, un-pinned dependency
`git+https://github.com/openai/CLIP.git` (requirements.txt:2).  The wrappers
(prefix splice, EOT gather, UPT mixer, leaderboard) ARE pinned: oracle/gen_golden.py
imports the reference's own models/*.py and utils/clip_pseudolabels.py
unmodified (this container only) and drives them on top of oracle/clip, and the
outputs are committed under tests/golden/.  The transformer arithmetic under the
wrappers is restated from the published openai/CLIP architecture and
cross-checked against the independent `transformers.CLIPModel` implementation
(tests/test_oracle_vs_hf.py): "parity unpinned" against openai/CLIP itself.
"""
