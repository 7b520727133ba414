"""`python -m particlesmc_amd params.toml` -- the reference's `particlesmc params.toml` on the HIP path."""
import sys

from .cli import main

sys.exit(main())
