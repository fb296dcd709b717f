"""advmix_amd.core - host mirror of the reference's lib/core entry points on the MI355X path."""
