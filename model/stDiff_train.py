from stdiff_amd.model.stDiff_train import normal_train_stDiff  # noqa: F401  (reference: model/stDiff_train.py)
