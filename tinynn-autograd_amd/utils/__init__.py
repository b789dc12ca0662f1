"""Counterparts of the reference's `utils` package used on the hot path (iterator, seeder)."""
