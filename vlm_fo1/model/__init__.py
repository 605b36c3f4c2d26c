"""`vlm_fo1.model` of the MI355X engine: builder.load_pretrained_model + the engine-backed model class."""
