## R-side of the drop-in: the reference's signatures (R/plaid.R:60, :213, :244, :554, :589,
## :631 of bigomics/plaid) with bodies that hand the arithmetic to libplaidhip.so through
## .Call().  Gene-name alignment and dimnames stay in R; nothing numeric is computed here.
## A maintainer of the reference would add `useDynLib(plaidhip, .registration = TRUE)` to
## NAMESPACE and replace the bodies of the functions below (see INTEGRATION.md).

.stat_code <- function(stats) match(stats[1], c("mean", "sum")) - 1L
.ties_code <- function(ties.method) {
  code <- match(ties.method, c("average", "min", "max"))
  if (is.na(code)) stop("ties.method '", ties.method, "' is not built (average/min/max)")
  code - 1L
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
  if (inherits(X, "CsparseMatrix")) {
    S <- .Call("R_plaidhip_plaid_csc", X@p, X@i, as.double(X@x), nrow(X), pat$Gp, pat$Gi,
               .stat_code(stats), normalize, PACKAGE = "plaidhip")
  } else {
    storage.mode(X) <- "double"
    S <- .Call("R_plaidhip_plaid_dense", X, pat$Gp, pat$Gi, .stat_code(stats), normalize,
               PACKAGE = "plaidhip")
  }
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}

normalize_medians <- function(x, ignore.zero = NULL) {
  x <- as.matrix(x); storage.mode(x) <- "double"
  iz <- if (is.null(ignore.zero)) NA else as.logical(ignore.zero)
  out <- .Call("R_plaidhip_normalize_medians", x, iz, PACKAGE = "plaidhip")
  dimnames(out) <- dimnames(x)
  out
}

sparse_colranks <- function(X, signed = FALSE, ties.method = "average") {
  X <- methods::as(X, "CsparseMatrix")
  rX <- X
  rX@x <- .Call("R_plaidhip_colranks_csc", X@p, as.double(X@x), .ties_code(ties.method), signed,
                PACKAGE = "plaidhip")
  rX
}

colranks <- function(X, sparse = NULL, signed = FALSE, keep.zero = FALSE, ties.method = "average") {
  if (is.null(sparse)) sparse <- inherits(X, "CsparseMatrix")
  if (sparse && keep.zero) return(sparse_colranks(X, signed = signed, ties.method = ties.method))
  D <- as.matrix(X); storage.mode(D) <- "double"
  rX <- .Call("R_plaidhip_colranks_dense", D, .ties_code(ties.method), signed, PACKAGE = "plaidhip")
  dimnames(rX) <- dimnames(X)
  rX
}

replaid.sing <- function(X, matG) {
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  D <- as.matrix(X); storage.mode(D) <- "double"
  S <- .Call("R_plaidhip_sing_dense", D, pat$Gp, pat$Gi, PACKAGE = "plaidhip")
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}

replaid.ssgsea <- function(X, matG, alpha = 0) {
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  if (inherits(X, "CsparseMatrix")) {
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
  xa <- .x_args(X)
  S <- .Call("R_plaidhip_ucell", xa[[1]], xa[[2]], xa[[3]], nrow(X), ncol(X), pat$Gp, pat$Gi,
             as.double(Matrix::colSums(matG != 0)), as.double(rmax), PACKAGE = "plaidhip")
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}

replaid.aucell <- function(X, matG, aucMaxRank = ceiling(0.05 * nrow(X))) {
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  xa <- .x_args(X)
  S <- .Call("R_plaidhip_aucell", xa[[1]], xa[[2]], xa[[3]], nrow(X), ncol(X), pat$Gp, pat$Gi,
             as.double(aucMaxRank), PACKAGE = "plaidhip")
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}

replaid.scse <- function(X, matG, removeLog2 = NULL, scoreMean = FALSE) {
  pat <- .aligned_pattern(X, matG)
  if (is.null(pat)) { message("[plaid] ERROR. No overlapping features."); return(NULL) }
  xa <- .x_args(X)
  S <- .Call("R_plaidhip_scse", xa[[1]], xa[[2]], xa[[3]], nrow(X), ncol(X), pat$Gp, pat$Gi,
             if (is.null(removeLog2)) NA else as.logical(removeLog2), as.logical(scoreMean),
             PACKAGE = "plaidhip")
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
plaid.test <- function(X, y, G, gsetX = NULL, tests = c("one", "two", "lm"),
                       metap.method = "fisher", sort.by = "p.meta") {
  if (!all(unique(y) %in% c(0, 1))) stop("elements of y must be 0 or 1")
  if (is.list(G)) {
    message("[plaid.test] converting gmt to sparse matrix...")
    G <- gmt2mat(G)
  }
  if (!metap.method %in% c("fisher", "sumlog", "stouffer", "sumz")) stop("Invalid method: ", metap.method)
  gg <- intersect(rownames(G), rownames(X))
  X <- as.matrix(X[gg, , drop = FALSE])
  G <- methods::as(G[gg, , drop = FALSE] != 0, "dgCMatrix")
  if (!is.null(gsetX)) gsetX <- as.matrix(gsetX[colnames(G), , drop = FALSE])
  bits <- sum(c(one = 1L, two = 2L, lm = 4L)[intersect(tests, c("one", "two", "lm"))])
  r <- .Call("R_plaidhip_plaid_test", X, as.integer(y), G@p, G@i, gsetX, bits,
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
  S <- .Call("R_plaidhip_gsva", as.matrix(X), pat$Gp, pat$Gi, as.numeric(tau), as.integer(rowtf == "ecdf"), PACKAGE = "plaidhip")
  dimnames(S) <- list(colnames(matG), colnames(X))
  S
}
