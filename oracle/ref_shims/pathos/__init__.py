"""Stand-in for pathos (process pool over walkers)."""
