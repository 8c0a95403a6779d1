from stdiff_amd.model.sample import model_sample_stDiff, sample_stDiff  # noqa: F401  (reference: model/sample.py)
