"""Device-backed counterparts of the reference's `core` package (same module names and public API)."""
