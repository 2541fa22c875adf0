"""mavflow -- host-side mirror of the reference's per-frame-pair interface over libmavflow.so (HIP, gfx950)."""
