"""``Model`` carrier handed to ``Move.propose`` (same fields as eryn/model.py:8-18)."""
from collections import namedtuple

Model = namedtuple("Model", ("log_like_fn", "compute_log_like_fn", "compute_log_prior_fn",
                             "temperature_control", "map_fn", "random"))
