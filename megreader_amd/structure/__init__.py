"""GPU mirrors of the reference's evaluation-side structure classes (representers / measurers)."""
