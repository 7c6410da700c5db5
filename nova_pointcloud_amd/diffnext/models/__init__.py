"""Models (reference diffnext/models)."""
