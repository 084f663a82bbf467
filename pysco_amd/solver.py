"""Particle-mesh force solve on the GPU: drop-in for the reference's solver.pm().

Same signature, option strings, `param` side effects and exceptions as reference solver.py:30-522;
every O(N^3)/O(Np) stage is a HIP kernel or a rocFFT plan behind libpmstep.so.  `position`,
`potential` and `additional_field` may be NumPy arrays (uploaded each call: correct but
PCIe-bound) or torch CUDA tensors (device-resident across steps: the intended mode).
"""
from __future__ import annotations

import logging
from typing import List

import numpy as np

from . import _lib as L
from . import fourier, iostream, mesh, utils

_C_LIGHT = 299792458.0  # astropy.constants.c (solver.py:14)
_WRITE_PK = True       # the slab layer lets only rank 0 write the P(k) file
_FUSE_GRADIENT = True  # tests flip this to compare with the unfused derivative + interpolation
_USE_FUSED_FFT = True  # False: rocFFT r2c -> Green's function -> c2r (always used when P(k) is written)


def _empty(template):
    if L.is_device_array(template):
        torch = L.require_gpu()
        return torch.empty(0, dtype=torch.float32, device=template.device)
    return np.empty(0, dtype=np.float32)


def pm(position, param, potential=None, additional_field=None, tables: List = (), *, _kick=None, _density=None,
       _rebucket=None):
    """Acceleration at the particles, potential and (for f(R)/MOND) the extra field.

    Mirrors solver.py:30-215: deposit -> density scaling -> [P(k)] -> extra field -> rhs ->
    Poisson solve (fft | fft_7pt | multigrid) -> gradient -> inverse interpolation.

    `_kick=(velocity, half_dt)` is the private hook integration.leapfrog uses to fuse its second
    kick (integration.py:262) into the interpolation kernel, and `_density` the density it already
    deposited together with its first kick and drift; neither is part of the reference API.
    `_rebucket` (a dict, private): ask the gather to write the particle arrays out of place in tile order
    (mesh.derivative_interp_vec_kick_rebucket); when it did, the dict holds the new "position", "velocity" (and
    "ids" if it came with some) and the returned acceleration is in the new order.
    """
    from . import real64, slab

    if real64.is_float64(position):  # the fp64 device mode (Newtonian FFT step): pm_real64_* entry points
        if slab.active():
            raise NotImplementedError("the fp64 device mode is single-GPU")
        return real64.pm(position, param, potential, additional_field, tables, _kick=_kick)
    if slab.active():  # one process per GPU: x-slab decomposition (no reference counterpart)
        return slab.get(2 ** int(param["ncoarse"])).pm(position, param, potential, additional_field, tables,
                                                       kick=_kick)
    L.require_gpu()
    if potential is None:
        potential = _empty(position)
    if additional_field is None:
        additional_field = _empty(position)
    ncells_1d = 2 ** int(param["ncoarse"])
    MASS_SCHEME = param["mass_scheme"].casefold()
    THEORY = param["theory"].casefold()
    pos_d = _dev(position)

    # (NGP -- mesh.py:2240-2278, a utility in the reference, which refuses it here, solver.py:80-93 -- runs through the
    # same fused kernels as an extension the caller has to ask for: param["allow_ngp"]; window exponent 1)
    if MASS_SCHEME not in ("cic", "tsc") and not (MASS_SCHEME == "ngp" and param.get("allow_ngp", False)):
        raise NotImplementedError(f"{param['mass_scheme']=}, should be 'CIC' or 'TSC'")
    param["MAS_index"] = {"ngp": 1, "cic": 2, "tsc": 3}[MASS_SCHEME]
    ctx = L.context(ncells_1d)
    # tile ranges + straggler lists once for the deposit and the gather of these positions (pmstep.h)
    ctx.call("pm_tile_lists", L.ptr(pos_d), pos_d.shape[0])
    try:
        # TSC == TSC_seq here (solver.py:86-89)
        khalo = None
        if _density is not None:
            # integration.leapfrog hands over the density its fused kick + drift + deposit made, possibly with
            # the tiles' k-halo contributions still in a side buffer (pmstep.h: pm_kick_drift_wrap_deposit_khalo)
            density, khalo = _density if isinstance(_density, tuple) else (_density, None)
        else:
            density = {"ngp": mesh.NGP, "cic": mesh.CIC, "tsc": mesh.TSC}[MASS_SCHEME](pos_d, ncells_1d)

        # Without an f(R) term the force grid is only ever read by the gather: the gather kernel then takes
        # the potential and differentiates its own LDS tile (mesh.derivative_interp_vec_kick), and the
        # [n,n,n,3] array is never built.
        fuse_gradient = (_FUSE_GRADIENT and ncells_1d >= 16 and THEORY != "fr"
                         and param["linear_newton_solver"].casefold() != "full_fft")  # (its force is not a gradient)
        # More stragglers than the lists hold (an unsorted or long-unsorted stream): the gather from the
        # potential would serve each of them with 324 loads; the force grid + order-independent gather it is.
        frac, lists_complete = utils.straggler_fraction(ncells_1d)
        fuse_gradient = fuse_gradient and lists_complete
        potential_d, additional_field_d, force = _solve_grid(density, param, _dev(potential),
                                                             _dev(additional_field), tables,
                                                             want_force=not fuse_gradient, khalo=khalo)
        del density

        if fuse_gradient:
            vel, half_dt = (None, 0.0) if _kick is None else _kick
            if (_rebucket is not None and _rebucket.get("mode") == "codes" and vel is not None
                    and L.is_device_array(position)):
                acceleration = mesh.derivative_interp_vec_kick_codes(MASS_SCHEME, potential_d, param["gradient_stencil_order"],
                                                                     pos_d, vel, half_dt)
                _rebucket["done"] = True
            elif _rebucket is not None and vel is not None and L.is_device_array(position):
                acceleration, _rebucket["position"], _rebucket["velocity"], _rebucket["ids"] = (
                    mesh.derivative_interp_vec_kick_rebucket(MASS_SCHEME, potential_d, param["gradient_stencil_order"],
                                                             pos_d, vel, half_dt, _rebucket.get("ids")))
            else:
                acceleration = mesh.derivative_interp_vec_kick(MASS_SCHEME, potential_d,
                                                               param["gradient_stencil_order"], pos_d, vel, half_dt)
        elif _kick is not None and ncells_1d >= 16:
            if not lists_complete:
                ctx.call("pm_set_gather_mode", 2)  # chunk / anchor kernel: any particle order through LDS
            try:
                if (_rebucket is not None and _rebucket.get("mode") == "codes" and lists_complete
                        and L.is_device_array(position)):
                    acceleration = mesh.derivative_interp_vec_kick_codes(MASS_SCHEME, force, 0, pos_d, _kick[0], _kick[1])
                    _rebucket["done"] = True
                elif _rebucket is not None and lists_complete and L.is_device_array(position):
                    acceleration, _rebucket["position"], _rebucket["velocity"], _rebucket["ids"] = (
                        mesh.derivative_interp_vec_kick_rebucket(MASS_SCHEME, force, 0, pos_d, _kick[0], _kick[1],
                                                                 _rebucket.get("ids")))
                else:
                    acceleration = mesh.interp_vec_kick(MASS_SCHEME, force, pos_d, _kick[0], _kick[1])
            finally:
                ctx.call("pm_set_gather_mode", 1)
        else:
            acceleration = {"ngp": mesh.invNGP_vec, "cic": mesh.invCIC_vec, "tsc": mesh.invTSC_vec}[MASS_SCHEME](force, pos_d)
            if _kick is not None:
                utils.add_vector_scalar_inplace(_kick[0], acceleration, -_kick[1])
        del force
    finally:
        ctx.call("pm_tile_lists", None, 0)
    if L.is_device_array(position):
        return acceleration, potential_d, additional_field_d
    return acceleration.cpu().numpy(), potential_d.cpu().numpy(), additional_field_d.cpu().numpy()


def _solve_grid(density, param, potential_d, additional_field_d, tables, want_force=True, khalo=None):
    """Everything of pm() between the deposit and the interpolation, on a full [n,n,n] density grid:
    normalisation, [P(k)], extra field, rhs, Poisson solve, force.  Returns (potential, additional_field,
    force[n,n,n,3]).  Shared by the single-GPU pm() and the replicated grid solve of the slab layer
    (which clears _WRITE_PK on all ranks but one)."""
    ncells_1d = density.shape[0]
    THEORY = param["theory"].casefold()
    if THEORY == "parametrized":  # solver.py:95-110
        evolution_term = param["aexp"] ** (-3 * (1 + param["w0"] + param["wa"])) * np.exp(
            -3 * param["wa"] * (1 - param["aexp"])
        )
        omega_lambda_z = (
            param["Om_lambda"] * evolution_term
            / (param["Om_m"] * param["aexp"] ** (-3) + param["Om_r"] * param["aexp"] ** (-4)
               + param["Om_lambda"] * evolution_term)
        )
        param["parametrized_mu_z"] = np.float32(1 + param["parametrized_mu0"] * omega_lambda_z / param["Om_lambda"])
    else:
        param["parametrized_mu_z"] = np.float32(1)

    # A k-halo side buffer is consumed by the fused FFT solve of the Newtonian path only (its z pass adds it
    # while loading the density); every other consumer gets the completed density.
    if khalo is not None and not (THEORY in ("newton", "parametrized") and _USE_FUSED_FFT and 32 <= ncells_1d <= 1024
                                  and param["linear_newton_solver"].casefold() in ("fft", "fft_7pt")):
        L.context(ncells_1d).call("pm_khalo_apply", L.ptr(density), L.ptr(khalo))
        khalo = None
    if ncells_1d**3 != param["npart"]:
        utils.prod_vector_scalar_inplace(density, np.float32(ncells_1d**3 / param["npart"]))
        if khalo is not None:
            utils.prod_vector_scalar_inplace(khalo, np.float32(ncells_1d**3 / param["npart"]))

    SAVE_POWER_SPECTRUM = param["save_power_spectrum"].casefold()
    if SAVE_POWER_SPECTRUM == "yes":
        param["save_pk"] = True
    elif SAVE_POWER_SPECTRUM == "z_out":
        param["save_pk"] = bool(param["write_snapshot"])
    elif SAVE_POWER_SPECTRUM == "no":
        param["save_pk"] = False
    else:
        raise NotImplementedError(f"{SAVE_POWER_SPECTRUM=}, should be 'yes', 'z_out' or 'no'")

    LINEAR_NEWTON_SOLVER = param["linear_newton_solver"].casefold()
    if khalo is not None and param["save_pk"]:  # the P(k) path transforms the density with rocFFT
        L.context(ncells_1d).call("pm_khalo_apply", L.ptr(density), L.ptr(khalo))
        khalo = None
    if param["save_pk"] and "multigrid" == LINEAR_NEWTON_SOLVER:  # solver.py:134-140
        density_fourier = fourier.fft_3D_real(density)
        k, Pk, Nmodes = fourier.fourier_grid_to_Pk(density_fourier, param["MAS_index"])
        del density_fourier
        Pk *= np.float32((param["boxlen"] / ncells_1d**2) ** 3)
        k *= np.float32(2 * np.pi / param["boxlen"])
        if _WRITE_PK:
            iostream.write_power_spectrum_to_ascii_file(k, Pk, Nmodes, param)

    param["compute_additional_field"] = True
    additional_field_d = get_additional_field(additional_field_d, density, param, tables)
    param["compute_additional_field"] = False
    # Newtonian FFT solve without P(k): rhs = f1*(rho - 1) is linear and its k = 0 mode is discarded by
    # the Green's function (fourier.py:491, 544, 595), so the scaling pass (solver.py:445-449) folds
    # into the Green's function prefactor: phi_k = (f1 * G) * rho_k.  One full grid pass less.
    fold_scale = (THEORY in ("newton", "parametrized") and LINEAR_NEWTON_SOLVER in ("fft", "fft_7pt")
                  and not param["save_pk"])
    if not fold_scale:
        rhs_poisson(density, additional_field_d, param)
    rhs = density

    if LINEAR_NEWTON_SOLVER == "multigrid":
        potential_d = initialise_potential(potential_d, rhs, param, tables)
        from . import multigrid

        potential_d = multigrid.linear(potential_d, rhs, param)
    elif LINEAR_NEWTON_SOLVER in ("fft", "fft_7pt"):
        if fold_scale:
            f1 = np.float32(1.5 * param["aexp"] * param["Om_m"] * param["parametrized_mu_z"])
            potential_d = fft(rhs, param, _source_scale=f1, _khalo=khalo)
        else:
            potential_d = fft(rhs, param)
    elif LINEAR_NEWTON_SOLVER == "full_fft":
        # The reference passes three arguments to the two-argument fft_force (solver.py:181, 200 vs :526-529) and
        # raises TypeError: that is what this does too unless the caller asks for the working path with
        # param["full_fft_force"] = True -- the branch as it spells itself out: the force comes from fft_force
        # (below) and the potential is left ALONE, i.e. it is whatever came in (possibly empty: do not switch to a
        # solver that warm-starts from it in mid-run).
        if not param.get("full_fft_force", False):
            raise TypeError("fft_force() takes 2 positional arguments but 3 were given")
    else:
        raise NotImplementedError(
            f"{param['linear_newton_solver']=}, should be multigrid, fft, fft_7pt or full_fft"
        )
    if LINEAR_NEWTON_SOLVER == "full_fft":
        force = fft_force(rhs, param)
        if THEORY == "fr":  # solver.py:181-189 (mesh.add_derivative_fR): force += half_c2 grad(u^(n+1))
            torch = L.require_gpu()
            force += mesh.derivative_fR(torch.zeros_like(additional_field_d), additional_field_d, fr_half_c2(param),
                                        param["fR_n"], param["gradient_stencil_order"])
        return potential_d, additional_field_d, force
    del rhs

    if THEORY == "fr":
        half_c2 = fr_half_c2(param)
        force = mesh.derivative_fR(potential_d, additional_field_d, half_c2, param["fR_n"],
                                   param["gradient_stencil_order"])
    elif want_force:
        force = mesh.derivative(potential_d, param["gradient_stencil_order"])
    else:
        if param["gradient_stencil_order"] not in (2, 3, 5, 7):  # same error as mesh.derivative
            raise ValueError(f"gradient_order={param['gradient_stencil_order']}, should be 2, 3, 5 or 7")
        force = None

    return potential_d, additional_field_d, force


def _dev(x):
    """NumPy or torch -> CUDA float32 tensor (identity for device tensors)."""
    torch = L.require_gpu()
    return L.to_device(x, torch.float32)


def fr_half_c2(param) -> np.float32:
    """0.5 * |f_R(a)| * c^2 in box units (solver.py:167-179)."""
    Rbar = 3 * param["Om_m"] * param["aexp"] ** (-3) + 12 * param["Om_lambda"]
    Rbar0 = 3 * param["Om_m"] + 12 * param["Om_lambda"]
    fR_a = -param["aexp"] ** 2 * ((Rbar0 / Rbar) ** (param["fR_n"] + 1)) * 10.0 ** (-param["fR_logfR0"])
    return np.float32(0.5 * (-fR_a) * (_C_LIGHT * 1e-3 * param["unit_t"] / (param["unit_l"] * param["aexp"])) ** 2)


def initialise_potential(potential, rhs, param, tables):
    """First guess for the (linear or scalaron) solve -- solver.py:218-282."""
    from . import cubic, laplacian

    if len(potential) == 0:
        if param["compute_additional_field"] and "fr" == param["theory"].casefold():
            q = param["fR_q"]
            if param["fR_n"] == 1:
                potential = cubic.initialise_potential(rhs, q)
            elif param["fR_n"] == 2:
                from . import quartic

                potential = quartic.initialise_potential(rhs, q)
            else:
                raise NotImplementedError(f"Only f(R) with n = 1 and 2, currently {param['fR_n']=}")
        else:
            potential = laplacian.initialise_potential(rhs)
    else:
        if not param["compute_additional_field"]:
            scaling = (
                param["aexp"] * utils.table_eval(tables[3], np.log(param["aexp"]))
                / (param["aexp_old"] * utils.table_eval(tables[3], np.log(param["aexp_old"])))
            )
            utils.prod_vector_scalar_inplace(potential, scaling)
    return potential


def get_additional_field(additional_field, density, param, tables):
    """Scalaron (f(R)) or Newtonian potential (MOND) -- solver.py:285-378."""
    THEORY = param["theory"].casefold()
    if THEORY in ("newton", "parametrized"):
        return _empty(density)
    if THEORY == "fr":
        from . import multigrid

        Rbar = 3 * param["Om_m"] * param["aexp"] ** (-3) + 12 * param["Om_lambda"]
        Rbar0 = 3 * param["Om_m"] + 12 * param["Om_lambda"]
        fR_a = -param["aexp"] ** 2 * ((Rbar0 / Rbar) ** (param["fR_n"] + 1)) * 10.0 ** (-param["fR_logfR0"])
        c2 = (_C_LIGHT * 1e-3 * param["unit_t"] / (param["unit_l"] * param["aexp"])) ** 2
        f1 = np.float32(param["aexp"] * param["Om_m"] / (c2 * 6)) / (-fR_a)
        f2 = np.float32(Rbar / 3 * param["aexp"] ** 4 - param["Om_m"] * param["aexp"]) / (6 * c2) / (-fR_a)
        dens_term = utils.linear_operator(density, f1, f2)
        q = np.float32(-param["aexp"] ** 4 * Rbar / (18 * c2)) / (-fR_a)
        param["fR_q"] = q
        additional_field = initialise_potential(additional_field, dens_term, param, tables)
        u_scalaron = multigrid.FAS(additional_field, dens_term, param)
        logging.info(f"{fR_a=}")
        return u_scalaron
    if THEORY == "mond":
        from . import multigrid

        rhs_poisson(density, additional_field, param)
        LINEAR_NEWTON_SOLVER = param["linear_newton_solver"].casefold()
        if LINEAR_NEWTON_SOLVER == "multigrid":
            additional_field = initialise_potential(additional_field, density, param, tables)
            additional_field = multigrid.linear(additional_field, density, param)
        elif LINEAR_NEWTON_SOLVER == "fft_7pt":
            additional_field = fft(density, param)
        else:
            raise NotImplementedError(
                f"{param['linear_newton_solver']=}, should be 'multigrid' or 'fft_7pt'"
            )
        return additional_field
    raise NotImplementedError(f"{param['theory']=}, should be 'newton', 'fr', 'parametrized' or 'mond'")


def rhs_poisson(density, additional_field, param) -> None:
    """In place: density -> right-hand side of the Poisson equation (solver.py:381-449)."""
    compute_MOND_potential = param["compute_additional_field"] is False and "mond" == param["theory"].casefold()
    if compute_MOND_potential:
        from . import mond

        g0 = (
            param["mond_g0"] * 1e-3 * 1e-10 * param["unit_t"] ** 2 / param["unit_l"]
            * param["aexp"] ** (1 + param["mond_scale_factor_exponent"])
        )
        alpha = param["mond_alpha"]
        MOND_FUNCTION = param["mond_function"].casefold()
        if MOND_FUNCTION == "simple":
            mond.rhs_simple(additional_field, density, g0)
        elif MOND_FUNCTION == "n":
            mond.rhs_n(additional_field, density, g0, n=alpha)
        elif MOND_FUNCTION == "beta":
            mond.rhs_beta(additional_field, density, g0, beta=alpha)
        elif MOND_FUNCTION == "gamma":
            mond.rhs_gamma(additional_field, density, g0, gamma=alpha)
        elif MOND_FUNCTION == "delta":
            mond.rhs_delta(additional_field, density, g0, delta=alpha)
        else:
            raise NotImplementedError(f"{MOND_FUNCTION=}, should be 'simple', 'n', 'beta', 'gamma' or 'delta'")
    else:
        f1 = np.float32(1.5 * param["aexp"] * param["Om_m"] * param["parametrized_mu_z"])
        utils.linear_operator_inplace(density, f1, -f1)


def fft_force(rhs, param):
    """Force grid [n, n, n, 3] of the `full_fft` solver (solver.py:524-566): r2c -> [P(k)] ->
    gradient_inverse_laplacian(_compensated) -> three c2r transforms."""
    torch = L.require_gpu()
    MAS_index = param["MAS_index"]
    rhs_d = _dev(rhs)
    n = rhs_d.shape[0]
    rhs_fourier = torch.empty((n, n, n // 2 + 1), dtype=torch.complex64, device=rhs_d.device)
    L.context(n).call("pm_fft_r2c", L.ptr(rhs_d), L.ptr(rhs_fourier))
    planar = fourier._grad_planar(rhs_fourier, 0 if MAS_index == 0 else int(MAS_index))
    if param.get("save_pk", False):
        k, Pk, Nmodes = fourier.fourier_grid_to_Pk(rhs_fourier, MAS_index)
        Pk *= np.float32((param["boxlen"] / n**2) ** 3 / (1.5 * param["aexp"] * param["Om_m"]) ** 2
                         / param["parametrized_mu_z"] ** 2)
        k *= np.float32(2 * np.pi / param["boxlen"])
        if _WRITE_PK:
            iostream.write_power_spectrum_to_ascii_file(k, Pk, Nmodes, param)
    del rhs_fourier
    return L.like_input(fourier._ifft_grad_planar(planar), rhs)


def fft(rhs, param, _source_scale=None, _khalo=None):
    """FFT Poisson solve (solver.py:453-522): r2c -> [P(k)] -> Green's function -> c2r.
    `_source_scale` (private) multiplies the Green's function: see the fold in pm()."""
    torch = L.require_gpu()
    MAS_index = param["MAS_index"]
    LINEAR_NEWTON_SOLVER = param["linear_newton_solver"].casefold()
    compute_MOND_potential = param["compute_additional_field"] is False and param["theory"] == "mond".casefold()
    if LINEAR_NEWTON_SOLVER == "fft":
        kind, p = (L.GREEN_PLAIN, 0) if MAS_index == 0 else (L.GREEN_COMPENSATED, int(MAS_index))
    elif LINEAR_NEWTON_SOLVER == "fft_7pt":
        kind, p = L.GREEN_7PT, 0
    else:
        raise NotImplementedError(f"{LINEAR_NEWTON_SOLVER=}, should be 'fft' or 'fft_7pt'")
    rhs_d = _dev(rhs)
    n = rhs_d.shape[0]
    ctx = L.context(n)
    k2, s, c0 = fourier.green_tables(n, kind)
    if _source_scale is not None:
        c0 = float(np.float32(c0) * np.float32(_source_scale))
    want_pk = ("save_pk" in param) and param["save_pk"] and not compute_MOND_potential
    phi = torch.empty_like(rhs_d)
    if not want_pk and 32 <= n <= 1024 and _USE_FUSED_FFT:
        # five LDS-resident FFT passes with the Green's function inside the x pass (csrc/pm_fftpoisson.hip)
        if _khalo is not None:
            ctx.call("pm_poisson_fft_fused_khalo", kind, p, L.ptr(rhs_d), L.ptr(_khalo), L.ptr(phi), L.hptr(k2),
                     L.hptr(s), c0)
        else:
            ctx.call("pm_poisson_fft_fused", kind, p, L.ptr(rhs_d), L.ptr(phi), L.hptr(k2), L.hptr(s), c0)
        return phi
    if _khalo is not None:
        raise RuntimeError("a k-halo buffer reached the unfused FFT path (solver._solve_grid should have applied it)")
    # three C-ABI calls (== pm_poisson_fft) so that per-stage event timing sees each of them
    rhs_fourier = torch.empty((n, n, n // 2 + 1), dtype=torch.complex64, device=rhs_d.device)
    ctx.call("pm_fft_r2c", L.ptr(rhs_d), L.ptr(rhs_fourier))
    if want_pk:
        k, Pk, Nmodes = fourier.fourier_grid_to_Pk(rhs_fourier, MAS_index)
        Pk *= np.float32(
            (param["boxlen"] / n**2) ** 3 / (1.5 * param["aexp"] * param["Om_m"]) ** 2
            / param["parametrized_mu_z"] ** 2
        )
        k *= np.float32(2 * np.pi / param["boxlen"])
        if _WRITE_PK:
            iostream.write_power_spectrum_to_ascii_file(k, Pk, Nmodes, param)
    ctx.call("pm_greens", kind, p, L.ptr(rhs_fourier), L.hptr(k2), L.hptr(s), c0)
    ctx.call("pm_fft_c2r", L.ptr(rhs_fourier), L.ptr(phi))
    return phi
