"""Ladder sharding across GPUs (one process per GPU).  Filled in below."""
