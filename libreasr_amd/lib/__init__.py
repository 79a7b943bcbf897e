"""Host-side mirror of the reference's Python inference surface (libreasr/lib/*), backed by the
gfx950 engine.  Same names, argument meaning and return shapes as the reference so that
api-server.py's ASRServicer runs unchanged."""
