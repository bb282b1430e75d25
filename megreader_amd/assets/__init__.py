"""Mirror of the reference's `assets` namespace (only assets.ops.dcn is code; the rest of assets/ is data)."""
