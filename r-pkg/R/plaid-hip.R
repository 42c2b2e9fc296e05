## R-side of the drop-in: the reference's signatures (R/plaid.R:60, :213, :244, :554, :589,
## :631 of bigomics/plaid) with bodies that hand the arithmetic to libplaidhip.so through
## .Call().  Gene-name alignment and dimnames stay in R; nothing numeric is computed here.
## A maintainer of the reference would add `useDynLib(plaidhip, .registration = TRUE)` to
## NAMESPACE and replace the bodies of the functions below (see INTEGRATION.md).

## Session options (set with options(); read on every call):
##   plaidhip.device     integer, the GPU ordinal the session context lives on (default 0; read when the
##                       context is first created)
##   plaidhip.devices    integer vector of GPU ordinals: with more than one, plaid() / replaid.sing() /
##                       replaid.ssgsea() shard the sample columns over them (a host thread per device inside
##                       the library, no process per GPU); one ordinal: that GPU is the session device; default: the
##                       session device alone.  plaidhip.precision applies to every device of the list.
##   plaidhip.precision  "f64" (default: scores equal to the last bits) or "mixed" (dense crossprod stages the
##                       sample columns as fp32, sums fp64; ~1e-7 relative, inside the 1e-5 bar; ~1.5x faster)
## a single ordinal in plaidhip.devices IS the session device (it used to be ignored in favour of plaidhip.device)
.device <- function() {
  d <- getOption("plaidhip.devices", NULL)
  if (length(d) == 1L) return(as.integer(d))
  as.integer(getOption("plaidhip.device", 0L))
}
.devices <- function() as.integer(getOption("plaidhip.devices", .device()))
.session <- function() {
  prec <- match(getOption("plaidhip.precision", "f64"), c("f64", "mixed"))
  if (is.na(prec)) stop("options(plaidhip.precision) must be \"f64\" or \"mixed\"")
  .Call("R_plaidhip_session", .device(), prec - 1L, PACKAGE = "plaidhip")
  invisible(NULL)
}

.stat_code <- function(stats) match(stats[1], c("mean", "sum")) - 1L
## ties.method is passed through like the reference does (R/plaid.R:614-617, 639-642).  `allowed`: what the function
## behind the branch takes (match.arg there): matrixStats::colRanks all seven, base::rank no "dense",
## sparseMatrixStats::colRanks max / average / min.  "random" is refused by the library (not a function of the input).
.ties_all <- c("average", "min", "max", "first", "last", "dense", "random")
.ties_code <- function(ties.method, allowed = .ties_all) {
  ties.method <- match.arg(ties.method, allowed)
  match(ties.method, .ties_all) - 1L
}

## intersect + binarise (reference R/plaid.R:65-73) WITHOUT copying X: the membership pattern
## is re-indexed into X's row space; returns NULL when nothing overlaps.
.aligned_pattern <- function(X, matG) {
  G <- methods::as(methods::as(matG, "CsparseMatrix"), "generalMatrix")
  first <- !duplicated(rownames(G))
  to_x <- match(rownames(G), rownames(X))          # first match in X, NA if absent
  to_x[!first] <- NA
  if (all(is.na(to_x))) return(NULL)
  col <- rep.int(seq_len(ncol(G)), diff(G@p))
  new_i <- to_x[G@i + 1L]
  keep <- !is.na(new_i) & G@x != 0
  list(Gp = c(0L, cumsum(tabulate(col[keep], nbins = ncol(G)))),
       Gi = as.integer(new_i[keep] - 1L))
}

plaid <- function(X, matG, stats = c("mean", "sum"), chunk = NULL, normalize = TRUE) {
  stats <- stats[1]
  if (NCOL(X) == 1) X <- cbind(X)
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) {
    message("[plaid] ERROR. No overlapping features.")
    return(NULL)
  }
  .session()
  dev <- .devices()
  if (length(dev) > 1L) {                                  # sample shards over several GPUs of the node
    xa <- .x_args(X)
    S <- .Call("R_plaidhip_plaid_multi", dev, xa[[1]], xa[[2]], xa[[3]], nrow(X), ncol(X), pat$Gp, pat$Gi,
               .stat_code(stats), normalize, PACKAGE = "plaidhip")
  } else if (inherits(X, "CsparseMatrix")) {
    S <- .Call("R_plaidhip_plaid_csc", X@p, X@i, as.double(X@x), nrow(X), pat$Gp, pat$Gi,
               .stat_code(stats), normalize, PACKAGE = "plaidhip")
  } else {
    X <- as.matrix(X); storage.mode(X) <- "double"
    S <- .Call("R_plaidhip_plaid_dense", X, pat$Gp, pat$Gi, .stat_code(stats), normalize,
               PACKAGE = "plaidhip")
  }
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}

## chunked_crossprod(x, y, chunk), R/plaid.R:100-123: t(x) %*% y -- internal in the reference too.  For the
## membership matrix plaid() builds (0/1, optionally scaled per column, R/plaid.R:73-77) the column scale is read off x
## and applied to the device's unscaled sums (the scheduled membership kernels); an x whose stored values differ inside
## a column (weighted / signed sets) goes to the general sparse kernel with its @p / @i / @x slots as they are.  The
## chunk loop and its message are the reference's (the device has no 2^31 limit, the loop only bounds the size of one
## transfer).
chunked_crossprod <- function(x, y, chunk = NULL) {
  x <- methods::as(methods::as(x, "CsparseMatrix"), "generalMatrix")
  if (nrow(x) != nrow(y)) stop("non-conformable arguments")
  col <- rep.int(seq_len(ncol(x)), diff(x@p))
  nz <- x@x != 0 | is.na(x@x)
  lo <- tapply(x@x[nz], factor(col[nz], levels = seq_len(ncol(x))), min)
  hi <- tapply(x@x[nz], factor(col[nz], levels = seq_len(ncol(x))), max)
  weighted <- any(lo != hi, na.rm = TRUE) || any(!is.finite(x@x[nz]))
  if (is.null(chunk) || chunk < 0) chunk <- round(0.8 * 2147483647 / ncol(x))     # R/plaid.R:103-104
  .session()
  if (weighted) {
    Wx <- as.double(x@x)
    block <- function(jj) {
      yy <- y[, jj, drop = FALSE]
      if (inherits(yy, "CsparseMatrix")) {
        .Call("R_plaidhip_crossprod_weighted_csc", x@p, x@i, Wx, yy@p, yy@i, as.double(yy@x), nrow(yy),
              PACKAGE = "plaidhip")
      } else {
        yy <- as.matrix(yy); storage.mode(yy) <- "double"
        .Call("R_plaidhip_crossprod_weighted_dense", x@p, x@i, Wx, yy, PACKAGE = "plaidhip")
      }
    }
    scale <- 1
  } else {
    scale <- ifelse(is.na(lo), 1, lo)
    Gp <- c(0L, cumsum(tabulate(col[nz], nbins = ncol(x))))
    Gi <- x@i[nz]
    block <- function(jj) {
      yy <- y[, jj, drop = FALSE]
      if (inherits(yy, "CsparseMatrix")) {
        .Call("R_plaidhip_plaid_csc", yy@p, yy@i, as.double(yy@x), nrow(yy), Gp, Gi, 1L, FALSE, PACKAGE = "plaidhip")
      } else {
        yy <- as.matrix(yy); storage.mode(yy) <- "double"
        .Call("R_plaidhip_plaid_dense", yy, Gp, Gi, 1L, FALSE, PACKAGE = "plaidhip")
      }
    }
  }
  if (ncol(y) < chunk) {
    gsetX <- block(seq_len(ncol(y)))
  } else {
    message("[chunked_crossprod] chunked compute: chunk = ", chunk)                # R/plaid.R:109
    gsetX <- matrix(NA_real_, ncol(x), ncol(y))
    k <- ceiling(ncol(y) / chunk)
    for (i in seq_len(k)) {
      jj <- ((i - 1) * chunk + 1):min(ncol(y), i * chunk)
      gsetX[, jj] <- block(jj)
    }
  }
  gsetX <- gsetX * scale
  dimnames(gsetX) <- list(colnames(x), colnames(y))
  gsetX
}

normalize_medians <- function(x, ignore.zero = NULL) {
  .session()
  x <- as.matrix(x); storage.mode(x) <- "double"
  iz <- if (is.null(ignore.zero)) NA else as.logical(ignore.zero)
  out <- .Call("R_plaidhip_normalize_medians", x, iz, PACKAGE = "plaidhip")
  dimnames(out) <- dimnames(x)
  out
}

sparse_colranks <- function(X, signed = FALSE, ties.method = "average") {
  .session()
  X <- methods::as(X, "CsparseMatrix")
  rX <- X
  rX@x <- .Call("R_plaidhip_colranks_csc", X@p, as.double(X@x),
                .ties_code(ties.method, c("average", "first", "last", "random", "max", "min")), signed,
                PACKAGE = "plaidhip")
  rX
}

colranks <- function(X, sparse = NULL, signed = FALSE, keep.zero = FALSE, ties.method = "average") {
  if (is.null(sparse)) sparse <- inherits(X, "CsparseMatrix")
  if (sparse && keep.zero) return(sparse_colranks(X, signed = signed, ties.method = ties.method))
  .session()
  ## the `sparse` ARGUMENT picks the function (R/plaid.R:598-619): sparseMatrixStats::colRanks or matrixStats::colRanks
  code <- .ties_code(ties.method, if (sparse) c("max", "average", "min") else .ties_all)
  if (inherits(X, "CsparseMatrix") && code <= 2L) {
    ## zeros are ranked and the result is dense (sparseMatrixStats::colRanks, R/plaid.R:602-609), but the
    ## matrix goes to the device as its three CSC slots: no as.matrix(X) on the host
    rX <- .Call("R_plaidhip_colranks_csc_dense", X@p, X@i, as.double(X@x), nrow(X), code,
                signed, PACKAGE = "plaidhip")
  } else {
    D <- as.matrix(X); storage.mode(D) <- "double"
    rX <- .Call("R_plaidhip_colranks_dense", D, code, signed, PACKAGE = "plaidhip")
  }
  dimnames(rX) <- dimnames(X)
  rX
}

replaid.sing <- function(X, matG) {
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  .session()
  dev <- .devices()
  if (inherits(X, "CsparseMatrix")) {
    ## the reference densifies a sparse X to rank its zeros (R/plaid.R:602-609); here the slots go to the device(s)
    X <- methods::as(X, "generalMatrix")
    S <- if (length(dev) > 1L) .Call("R_plaidhip_sing_csc_multi", dev, X@p, X@i, as.double(X@x), nrow(X), pat$Gp, pat$Gi,
                                     PACKAGE = "plaidhip")
         else .Call("R_plaidhip_sing_csc", X@p, X@i, as.double(X@x), nrow(X), pat$Gp, pat$Gi, PACKAGE = "plaidhip")
  } else {
    D <- as.matrix(X); storage.mode(D) <- "double"
    S <- if (length(dev) > 1L) .Call("R_plaidhip_sing_multi", dev, D, pat$Gp, pat$Gi, PACKAGE = "plaidhip")
         else .Call("R_plaidhip_sing_dense", D, pat$Gp, pat$Gi, PACKAGE = "plaidhip")
  }
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}

replaid.ssgsea <- function(X, matG, alpha = 0) {
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  .session()
  dev <- .devices()
  if (length(dev) > 1L) {
    xa <- .x_args(X)
    S <- .Call("R_plaidhip_ssgsea_multi", dev, xa[[1]], xa[[2]], xa[[3]], nrow(X), ncol(X), pat$Gp, pat$Gi,
               as.double(alpha), PACKAGE = "plaidhip")
  } else if (inherits(X, "CsparseMatrix")) {
    S <- .Call("R_plaidhip_ssgsea_csc", X@p, X@i, as.double(X@x), nrow(X), pat$Gp, pat$Gi,
               as.double(alpha), PACKAGE = "plaidhip")
  } else {
    D <- as.matrix(X); storage.mode(D) <- "double"
    S <- .Call("R_plaidhip_ssgsea_dense", D, pat$Gp, pat$Gi, as.double(alpha), PACKAGE = "plaidhip")
  }
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}

## X as the argument triple the shim expects: (p, i, values) for a dgCMatrix, (NULL, NULL, matrix) otherwise
.x_args <- function(X) {
  if (inherits(X, "CsparseMatrix")) list(X@p, X@i, as.double(X@x))
  else { D <- as.matrix(X); storage.mode(D) <- "double"; list(NULL, NULL, D) }
}

replaid.ucell <- function(X, matG, rmax = 1500) {
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  .session()
  xa <- .x_args(X)
  S <- .Call("R_plaidhip_ucell", xa[[1]], xa[[2]], xa[[3]], nrow(X), ncol(X), pat$Gp, pat$Gi,
             as.double(Matrix::colSums(matG != 0)), as.double(rmax), PACKAGE = "plaidhip")
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}

replaid.aucell <- function(X, matG, aucMaxRank = ceiling(0.05 * nrow(X))) {
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  .session()
  xa <- .x_args(X)
  S <- .Call("R_plaidhip_aucell", xa[[1]], xa[[2]], xa[[3]], nrow(X), ncol(X), pat$Gp, pat$Gi,
             as.double(aucMaxRank), PACKAGE = "plaidhip")
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}

replaid.scse <- function(X, matG, removeLog2 = NULL, scoreMean = FALSE) {
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  .session()
  xa <- .x_args(X)
  S <- .Call("R_plaidhip_scse", xa[[1]], xa[[2]], xa[[3]], nrow(X), ncol(X), pat$Gp, pat$Gi,
             if (is.null(removeLog2)) NA else as.logical(removeLog2), as.logical(scoreMean),
             PACKAGE = "plaidhip")
  if (isTRUE(attr(S, "removedLog2")))   ## R/plaid.R:163-164 (the NULL case is decided on the device)
    message("[replaid.scse] Converting data to linear scale (removing log2)...")
  attr(S, "removedLog2") <- NULL
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}


## gmt2mat(read.gmt(file)) as ONE native call (the text never becomes an R list): same ordering rules
## as R/gmt-utils.R:19-66 (sets by size, duplicated names dropped, head(ntop), genes by frequency,
## head(max.genes), rows by decreasing row sum).  50k sets: ~2 s instead of ~50 s.
gmt2mat.file <- function(gmt.file, dir = NULL, add.source = FALSE, nrows = -1,
                         max.genes = -1, ntop = -1, sparse = TRUE, bg = NULL) {
  f0 <- gmt.file
  if (strtrim(gmt.file, 1) == "/") dir <- NULL
  if (!is.null(dir)) f0 <- paste(sub("/$", "", dir), "/", gmt.file, sep = "")
  r <- .Call("R_plaidhip_gmt2mat_file", f0, add.source, nrows, max.genes, ntop, bg, PACKAGE = "plaidhip")
  D <- methods::new("dgCMatrix", p = r[[1]], i = r[[2]], x = rep(1, length(r[[2]])), Dim = r[[3]],
                    Dimnames = list(r[[4]], r[[5]]))
  if (!sparse) D <- as.matrix(D)
  D
}


## plaid.test(), R/plaid.R:392-474: same arguments and result; the group means of X, Gt fc, Gt fc^2 and the
## per-set Welch statistics are reduced on the device (with gsetX = NULL the score matrix never leaves it),
## only O(sets) numbers come back.
plaid.test <- function(X, y, G, gsetX, tests = c("one", "two", "lm"),
                       metap.method = "fisher", sort.by = "p.meta") {
  ## (the reference's signature, R/plaid.R:392: gsetX has no default; left out, it is recomputed from X and G like NULL)
  if (missing(gsetX)) gsetX <- NULL
  if (!all(unique(y) %in% c(0, 1))) stop("elements of y must be 0 or 1")
  if (is.list(G)) {
    message("[plaid.test] converting gmt to sparse matrix...")
    G <- gmt2mat(G)
  }
  if (!metap.method %in% c("fisher", "sumlog", "stouffer", "sumz")) stop("Invalid method: ", metap.method)
  gg <- intersect(rownames(G), rownames(X))
  X <- as.matrix(X[gg, , drop = FALSE]); storage.mode(X) <- "double"
  G <- G[gg, , drop = FALSE]
  pat <- .aligned_pattern(X, G)                           # stored zeros of G are dropped: set sizes count members only
  if (!is.null(gsetX)) { gsetX <- as.matrix(gsetX[colnames(G), , drop = FALSE]); storage.mode(gsetX) <- "double" }
  bits <- sum(c(one = 1L, two = 2L, lm = 4L)[intersect(tests, c("one", "two", "lm"))])
  .session()
  r <- .Call("R_plaidhip_plaid_test", X, as.integer(y), pat$Gp, pat$Gi, gsetX, bits,
             as.integer(metap.method %in% c("stouffer", "sumz")), PACKAGE = "plaidhip")
  keep <- c(TRUE, "one" %in% tests, "two" %in% tests, "lm" %in% tests, TRUE, TRUE)
  res <- r[, keep, drop = FALSE]
  dimnames(res) <- list(colnames(G), c("gsetFC", "p.one", "p.two", "p.lm", "p.meta", "q.meta")[keep])
  if (sort.by %in% colnames(res)) res <- res[order(res[, sort.by]), ]
  res
}


## replaid.gsva(), R/plaid.R:338-363: the row transform ("z" or "ecdf"), the signed ranks, the power and
## plaid() run on the device in one call.
replaid.gsva <- function(X, matG, tau = 0, rowtf = c("z", "ecdf")[1]) {
  rowtf <- rowtf[1]
  if (!rowtf %in% c("z", "ecdf")) stop("Error: unknown row transform", rowtf)
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  .session()
  D <- as.matrix(X); storage.mode(D) <- "double"
  S <- .Call("R_plaidhip_gsva", D, pat$Gp, pat$Gi, as.numeric(tau), as.integer(rowtf == "ecdf"), PACKAGE = "plaidhip")
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}
