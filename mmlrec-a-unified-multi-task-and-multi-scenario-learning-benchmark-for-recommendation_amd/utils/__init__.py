"""Host-side data utilities (counterpart of the reference's utils/)."""
