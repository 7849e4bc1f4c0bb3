"""empty stub: cv2 is only needed by the reference disk loader, not by the hot path."""
