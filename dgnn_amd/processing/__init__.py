"""Device-side counterparts of the reference's processing/ steps that sit directly before and after the hot path."""
