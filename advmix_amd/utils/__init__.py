"""advmix_amd.utils - host mirror of the reference's lib/utils entry points on the MI355X path."""
