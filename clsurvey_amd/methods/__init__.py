"""Host-side mirror of the reference's per-method trainers (src/methods/<M>/) on the HIP path."""
