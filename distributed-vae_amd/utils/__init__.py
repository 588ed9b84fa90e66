"""Host-side mirrors of ``mmidas/utils`` that belong to the hot path's data side."""
