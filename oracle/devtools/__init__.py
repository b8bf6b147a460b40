"""Developer diagnostics that use the oracle as the checker (test infrastructure, like everything under oracle/)."""
