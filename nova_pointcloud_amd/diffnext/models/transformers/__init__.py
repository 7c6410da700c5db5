"""Transformer generators (reference diffnext/models/transformers)."""
