"""Embedding tricks of the reference (tricks/): only the QR operator is on the scope table."""
