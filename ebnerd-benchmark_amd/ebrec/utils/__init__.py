"""Host-side data utilities of the NRMS path (pandas / numpy; polars frames are accepted and converted)."""
