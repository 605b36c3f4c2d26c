"""Drop-in `vlm_fo1` surface of the MI355X engine.

Same import paths and call signatures as om-ai-lab/VLM-FO1 (`vlm_fo1.model.builder.load_pretrained_model`,
`vlm_fo1.mm_utils.*`, `vlm_fo1.task_templates.*`, `vlm_fo1.constants.*`), so the reference's `inference.py`
and `evaluation/*` run unmodified against this package; every tensor op underneath is a libfo1hip.so kernel
(vlm_fo1_amd/)."""
