"""Clip file loading (mirror of track_mjx/io)."""
from .load import (generate_train_test_split, load_clips_metadata, load_data, load_reference_clip_data,  # noqa: F401
                   make_multiclip_data, make_singleclip_data, select_clips, sub_sample_training_set)
