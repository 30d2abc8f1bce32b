"""The name the snapshot's file carries (`_deprecated_contrastive.py`); same module as `contrastive`."""
from big_vision_amd.trainers.proj.image_text.contrastive import *  # noqa: F401,F403
from big_vision_amd.trainers.proj.image_text.contrastive import (  # noqa: F401
    loss_fn, make_update_fn, make_train_state, make_predict_fn, check_finite, get_model, LOSSES)
