"""Contributed layers on top of the hot path (mirrors the reference's `ratinabox.contribs`)."""
